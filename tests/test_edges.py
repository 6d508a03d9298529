"""Edge classes used stand-alone, following the checks of the reference's rectipy_tests/test_edges.py
(`Linear`: orientation rule, dtype handling, trainable-parameter bookkeeping, error behaviour; `RLS`: construction,
forward, update, argument validation).  `Linear` is plain torch and runs anywhere; `RLS.update` goes through the RLS
kernel (oracle on the CPU, HIP on the GPU box)."""
import numpy as np
import pytest
import torch

from rectipy_amd.edges import Linear
from rectipy_amd.rls import RLS

TARGETS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture
def where(request, oracle_backend):
    if request.param == "oracle":
        return "cpu", oracle_backend
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return "cuda:0", None


def test_linear_edge_like_reference_test_1_1():
    n, m = 10, 2
    rng = np.random.default_rng(0)
    w1 = rng.standard_normal((n, m))
    x = torch.randn(n, dtype=torch.float64)
    torch_lin = torch.nn.Linear(in_features=n, out_features=m, bias=False, dtype=torch.float64)

    lin1 = Linear(n, m)
    lin2 = Linear(m, n)
    lin3 = Linear(n, m, weights=w1.T + 2.0, dtype=torch.float64)
    lin4 = Linear(n, m, weights=w1, dtype=torch.float32)
    lin5 = Linear(n, m, weights=w1, detach=False, dtype=torch.float64)
    lin6 = Linear(n, m, weights=torch_lin.weight, dtype=torch.float64)

    # weights are stored [n_out, n_in]; an (n_in, n_out) matrix is transposed (edges.py:22-26)
    assert lin2.weights.shape == w1.shape
    assert lin1.weights.shape[0] == lin2.weights.shape[1]
    assert float((lin5.weights.detach() - torch.tensor(w1.T)).abs().sum()) == pytest.approx(0.0, abs=1e-12)
    assert lin3.weights.dtype == torch.float64 and lin4.weights.dtype == torch.float32

    # only a non-detached edge exposes (and tracks) its weights as a parameter
    assert len(list(lin5.parameters())) - len(list(lin4.parameters())) == 1
    assert len(list(lin1.parameters())) == 0
    assert lin1.weights.requires_grad is False and lin5.weights.requires_grad is True

    # forward
    assert float((lin5.forward(x).detach() - lin3.forward(x)).abs().sum()) > 0.0
    assert float((torch_lin.forward(x) - lin6.forward(x)).abs().sum()) == pytest.approx(0.0, abs=1e-12)
    # a whole window at once (this build's extension): rows of X are time steps
    X = torch.randn(7, n, dtype=torch.float64)
    assert torch.allclose(lin3.forward(X), torch.stack([lin3.forward(r) for r in X]))

    with pytest.raises(ValueError):
        Linear(n, m, weights=rng.standard_normal((n + 1, m + 1)))
    with pytest.raises(RuntimeError):
        lin4.forward(x)              # float32 weights, float64 input
    with pytest.raises(RuntimeError):
        lin1.forward(torch.randn(m))  # wrong input width


def test_rls_construction_and_validation():
    n, m = 10, 2
    w1 = np.random.default_rng(5).standard_normal((n, m))
    rls1 = RLS(n, m)
    rls2 = RLS(n, m, weights=w1)
    assert isinstance(rls1, Linear)
    assert float((rls2.weights - torch.tensor(w1.T, dtype=torch.float32)).abs().sum()) == pytest.approx(0.0, abs=1e-6)
    assert rls1.P.shape == (n, n) and torch.equal(rls1.P, torch.eye(n))
    assert float(rls1.weights.abs().sum()) == 0.0            # zero initial read-out (edges.py:215)
    assert len(list(rls2.parameters())) == 0                 # trained by update(), never by the optimiser
    assert RLS(n, m, beta=0.5).beta == pytest.approx(2.0)    # stores the inverse forgetting rate (edges.py:220)
    for bad in (dict(alpha=-0.5), dict(beta=1.5), dict(beta=-0.1)):
        with pytest.raises(ValueError):
            RLS(n, m, **bad)
    with pytest.raises(ValueError):
        RLS(n + 1, m, weights=w1)


@pytest.mark.parametrize("where", TARGETS, indirect=True)
def test_rls_forward_and_update_like_reference_test_1_2(where):
    dev, backend = where
    n, m = 10, 2
    gen = torch.Generator().manual_seed(5)
    w1 = torch.randn(n, m, generator=gen).numpy()
    x = torch.randn(n, generator=gen).to(dev)
    y = torch.randn(m, generator=gen).to(dev)
    rls1 = RLS(n, m, backend=backend).to(dev)
    rls2 = RLS(n, m, weights=w1, backend=backend).to(dev)
    rls3 = RLS(n, m, weights=w1, beta=0.5, backend=backend).to(dev)
    rls4 = RLS(n, m, weights=w1, alpha=0.1, backend=backend).to(dev)

    r1_1, r1_2 = rls1.forward(x), rls1.forward(x)
    assert r1_1.shape[0] == m and torch.equal(r1_1, r1_2)    # forward does not change the edge
    for rls in (rls2, rls3, rls4):
        y_hat = rls.forward(x)
        rls.update(x, y, y_hat)
        err = (y - y_hat).double()
        assert float(rls.loss) == pytest.approx(float(err @ err), rel=1e-5)
    r2, r3, r4 = rls2.forward(x), rls3.forward(x), rls4.forward(x)
    assert float((r2 - r3).abs().sum()) > 0 and float((r3 - r4).abs().sum()) > 0   # beta and alpha matter

    # the update itself, against the formula of the reference (edges.py:227-234) in float64
    W = torch.tensor(w1.T, dtype=torch.float64)
    P = torch.eye(n, dtype=torch.float64)
    xd, yd = x.double().cpu(), y.double().cpu()
    z = P @ xd
    k = 1.0 / (1.0 + xd @ z)
    W_ref = W + torch.outer(yd - k * xd @ (W + torch.outer(yd, z)).T, z)
    P_ref = P - k * torch.outer(z, z)
    assert (rls2.weights.double().cpu() - W_ref).abs().max() < 1e-5
    assert (rls2.P.double().cpu() - P_ref).abs().max() < 1e-5

    with pytest.raises(RuntimeError):
        rls1.forward(torch.randn(n + 1).to(dev))
    with pytest.raises(RuntimeError):
        rls1.update(x, torch.randn(m + 1).to(dev), torch.randn(m + 1).to(dev))
