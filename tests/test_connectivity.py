"""Host-side helpers exported for drop-in imports (rectipy/__init__.py:35-36)."""
import numpy as np
import pytest
from scipy.stats import rv_discrete

import rectipy_amd as ra


def test_connectivity_generators():
    np.random.seed(0)
    C = ra.random_connectivity(10, 20, 0.25)
    assert C.shape == (10, 20) and np.allclose(C.sum(1), 1) and ((C > 0).sum(1) == 5).all()
    assert set(np.unique(ra.random_connectivity(4, 8, 0.5, normalize=False))) == {0.0, 1.0}
    inp = ra.input_connections(30, 4, 0.5)
    assert inp.shape == (30, 4) and np.allclose(inp.sum(0), 0, atol=1e-12) and ((inp != 0).sum(0) == 15).all()
    d = rv_discrete(values=(np.arange(1, 6), np.ones(5) / 5))
    ring = ra.circular_connectivity(20, 0.3, d)
    assert np.allclose(ring.sum(1), 1) and ring.shape == (20, 20)
    line = ra.line_connectivity(20, 0.3, d, homogeneous_weights=False)
    assert line.shape == (20, 20) and (line >= 0).all()


def test_normalize_and_wta():
    x = np.random.default_rng(0).standard_normal((4, 5))
    ra.normalize(x, "zscore", row_wise=True)
    assert np.allclose(x.mean(1), 0, atol=1e-12) and np.allclose(x.std(1), 1)
    y = np.arange(6.0).reshape(2, 3)
    ra.normalize(y, "minmax")
    assert y.min() == 0 and y.max() == 1
    z = np.ones((2, 2))
    ra.normalize(z, "sum")
    assert z.sum() == pytest.approx(1.0)
    with pytest.raises(ValueError):
        ra.normalize(z, "nope")
    assert ra.wta_score(np.eye(3), np.eye(3)) == 1.0
    assert ra.wta_score(np.eye(3), np.roll(np.eye(3), 1, axis=1)) == 0.0
