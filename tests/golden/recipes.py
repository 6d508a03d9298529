"""Deterministic input recipes shared by the golden generators (build container) and the tests (everywhere).

The N >= 256 fixtures would be tens of MB if W and the (T, N) inputs were stored, so those arrays are
REGENERATED from a seed on both sides.  A library RNG is not safe for that (vectorised libm paths differ between
hosts), so the generator here is a counter-based integer hash: every value is an integer sum scaled by one IEEE
multiplication -- bit-identical on every machine.  Each fixture additionally stores `fingerprint()` of every
regenerated array, and the tests compare it before anything else.

Test infrastructure only (data recipes; nothing here is reference code).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _bits(count: int, seed: int) -> np.ndarray:
    idx = np.arange(count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return _mix(idx ^ _mix(np.full(count, seed, dtype=np.uint64)))


def uniform(shape, seed: int) -> np.ndarray:
    """float32 uniform on [0, 1): 24 hash bits / 2^24 (exact in float32)."""
    count = int(np.prod(shape))
    k = (_bits(count, seed) >> np.uint64(40)).astype(np.float64)
    return (k / 16777216.0).astype(np.float32).reshape(shape)


def normal(shape, seed: int) -> np.ndarray:
    """float32, approximately N(0, 1): Irwin-Hall sum of four 16-bit uniforms taken from one 64-bit hash, centred
    and scaled by sqrt(3)/2^16 (integer sum -> one float64 multiply -> one rounding to float32)."""
    count = int(np.prod(shape))
    b = _bits(count, seed)
    m = np.uint64(0xFFFF)
    s = ((b & m) + ((b >> np.uint64(16)) & m) + ((b >> np.uint64(32)) & m) + ((b >> np.uint64(48)) & m)).astype(np.float64)
    return ((s - 2.0 * 65535.0) * (np.sqrt(3.0) / 65536.0)).astype(np.float32).reshape(shape)


def fingerprint(a: np.ndarray) -> np.ndarray:
    """(sum, sum of squares, sum of index-weighted values) in float64 of the flattened array."""
    f = np.asarray(a, dtype=np.float64).reshape(-1)
    w = (np.arange(f.size, dtype=np.float64) % 9973.0) + 1.0
    return np.array([f.sum(), (f * f).sum(), (f * w).sum()])


# ---- the networks of the N >= 256 fixtures ----------------------------------------------------------------------

def li_network(n: int, seed: int, g: float, defaults: bool):
    """tanh leaky-integrator population.  defaults=True: the YAML defaults of SURVEY C1 (tau = 10, k = 1, eta = 0,
    y0 = 0; leaky_integrator.yaml:13-19); else per-neuron parameters as in the small fixtures."""
    W = (normal((n, n), seed) * np.float32(g / np.sqrt(n))).astype(np.float32)
    if defaults:
        tau, k, eta, y0 = (np.full(n, 10.0, np.float32), np.ones(n, np.float32), np.zeros(n, np.float32),
                           np.zeros(n, np.float32))
    else:
        tau = (5.0 + 5.0 * uniform(n, seed + 1)).astype(np.float32)
        k = (0.5 + uniform(n, seed + 2)).astype(np.float32)
        eta = (0.1 * normal(n, seed + 3)).astype(np.float32)
        y0 = normal(n, seed + 4)
    return dict(W=W, tau=tau, k=k, eta=eta, y0=y0)


def qif_network(n: int, seed: int, J: float = 20.0):
    """QIF population of SURVEY C4: W = rand * J / N, tau = k = tau_s = 1, eta = -5 + spread, v0 = -2, s0 = 0."""
    W = (uniform((n, n), seed) * np.float32(J / n)).astype(np.float32)
    eta = (-5.0 + normal(n, seed + 1)).astype(np.float32)
    one = np.ones(n, np.float32)
    y0 = np.concatenate([np.full(n, -2.0, np.float32), np.zeros(n, np.float32)])
    return dict(W=W, eta=eta, tau=one, k=one.copy(), tau_s=one.copy(), y0=y0)
