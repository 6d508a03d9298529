"""rp_expf / rp_tanhf (include/rectipy_math.h) against float64 libm: <= 2 ulp, and the row-dot summation
tree of the oracle against a float64 dot product."""
import ctypes as C

import numpy as np


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    ulp[ulp == 0] = np.finfo(np.float32).tiny
    return np.abs(got.astype(np.float64) - ref64) / ulp


def test_tanhf_accuracy(oracle_backend):
    lib = oracle_backend.lib
    xs = np.concatenate([np.linspace(-12, 12, 20001), np.linspace(-0.7, 0.7, 20001), [0.0, 1e-8, -1e-8, 0.625, -0.625,
                         30.0, -30.0]]).astype(np.float32)
    got = np.array([lib.oracle_tanhf(C.c_float(float(v))) for v in xs], dtype=np.float32)
    err = ulp_err(got, np.tanh(xs.astype(np.float64)))
    assert err.max() <= 2.0, f"tanhf max error {err.max()} ulp at x={xs[err.argmax()]}"
    assert lib.oracle_tanhf(C.c_float(float("inf"))) == 1.0 and lib.oracle_tanhf(C.c_float(float("-inf"))) == -1.0
    assert np.isnan(lib.oracle_tanhf(C.c_float(float("nan"))))


def test_expf_accuracy(oracle_backend):
    lib = oracle_backend.lib
    xs = np.linspace(-87, 88, 40001).astype(np.float32)
    got = np.array([lib.oracle_expf(C.c_float(float(v))) for v in xs], dtype=np.float32)
    err = ulp_err(got, np.exp(xs.astype(np.float64)))
    assert err.max() <= 2.0, f"expf max error {err.max()} ulp at x={xs[err.argmax()]}"
    assert lib.oracle_expf(C.c_float(100.0)) == float("inf") and lib.oracle_expf(C.c_float(-100.0)) == 0.0


def test_row_dot_tree(oracle_backend):
    lib = oracle_backend.lib
    rng = np.random.default_rng(0)
    for n, vec4 in ((10, 0), (64, 1), (250, 0), (1024, 1), (1000, 1), (4100, 1)):
        w = rng.standard_normal(n).astype(np.float32)
        r = rng.standard_normal(n).astype(np.float32)
        got = lib.oracle_row_dot(w.ctypes.data, r.ctypes.data, n, vec4, 8 * n + 5)
        ref = float(np.dot(w.astype(np.float64), r.astype(np.float64)))
        assert abs(got - ref) <= 1e-5 * np.sqrt(n)
