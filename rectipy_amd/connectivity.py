"""Host-side numpy helpers for building coupling matrices and scoring read-outs, offered under the names the
reference exports from ``rectipy`` (rectipy/__init__.py:35-36; semantics of rectipy/utility.py:76-276) so that
scripts importing them keep working.  They never touch the GPU path."""
import numpy as np


def _ring_or_line(N, p, spatial_distribution, homogeneous_weights, wrap):
    C = np.zeros((N, N))
    draws = int(N * p)
    for node in range(N):
        offsets = np.asarray(spatial_distribution.rvs(size=draws))
        direction = np.where(np.random.uniform(size=draws) < 0.5, -1, 1)
        sources = node + offsets * direction
        if wrap:
            sources = np.mod(sources, N)
        else:
            sources = sources[(sources > 0) & (sources < N)]
        uniq, counts = np.unique(sources, return_counts=True)
        if len(uniq) == 0:
            continue
        if homogeneous_weights:
            C[node, uniq] = 1.0 / len(uniq)
        else:
            C[node, uniq] = counts / (draws if wrap else len(sources))
    return C


def circular_connectivity(N: int, p: float, spatial_distribution, homogeneous_weights: bool = True) -> np.ndarray:
    """N x N coupling of nodes on a ring: each node draws int(N*p) source offsets from `spatial_distribution`
    (a scipy discrete distribution), sign chosen at random, indices wrapped around."""
    return _ring_or_line(N, p, spatial_distribution, homogeneous_weights, wrap=True)


def line_connectivity(N: int, p: float, spatial_distribution, homogeneous_weights: bool = True) -> np.ndarray:
    """As :func:`circular_connectivity` on an open line: draws that leave (0, N) are dropped."""
    return _ring_or_line(N, p, spatial_distribution, homogeneous_weights, wrap=False)


def random_connectivity(n: int, m: int, p: float, normalize: bool = True) -> np.ndarray:
    """n x m matrix with int(m*p) randomly placed entries per row (1/count each if `normalize`, else 1)."""
    C = np.zeros((n, m))
    count = int(m * p)
    if count == 0:
        return C
    cols = np.argsort(np.random.uniform(size=(n, m)), axis=1)[:, :count]
    np.put_along_axis(C, cols, 1.0 / count if normalize else 1.0, axis=1)
    return C


def input_connections(n: int, m: int, p: float, variance: float = 1.0, zero_mean: bool = True) -> np.ndarray:
    """n x m input weights: each column (input channel) projects to int(n*p) random rows with N(0, variance)-scaled
    weights; with `zero_mean` the weights of a column sum to zero."""
    support = random_connectivity(m, n, p, normalize=False).T > 0
    C = np.zeros((n, m))
    for col in range(m):
        rows = np.flatnonzero(support[:, col])
        w = np.random.randn(len(rows)) * variance
        if zero_mean and len(rows):
            w -= w.mean()
        C[rows, col] = w
    return C


def normalize(x: np.ndarray, mode: str = "minmax", row_wise: bool = False) -> np.ndarray:
    """In-place normalisation: 'minmax' (to [0, 1]), 'zscore', or 'sum' (to unit sum); optionally per row."""
    if row_wise:
        for i in range(x.shape[0]):
            normalize(x[i, :], mode=mode, row_wise=False)
        return x
    if mode == "minmax":
        x -= x.min()
        top = x.max()
        if top > 0:
            x /= top
    elif mode == "zscore":
        x -= x.mean()
        sd = x.std()
        if sd > 0:
            x /= sd
    elif mode == "sum":
        x /= x.sum()
    else:
        raise ValueError(f"Invalid normalization mode: {mode}.")
    return x


def wta_score(x: np.ndarray, y: np.ndarray) -> float:
    """Fraction of rows whose arg-max feature agrees between `x` and `y` (winner-takes-all score)."""
    return float(np.mean(np.argmax(x, axis=1) == np.argmax(y, axis=1)))
