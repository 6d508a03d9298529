"""ctypes binding of the CPU oracle (oracle/rnn_oracle.c) with the backend interface of
``rectipy_amd.engine`` -- TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import this.
It lets the host logic of rectipy_amd (sampling bookkeeping, facade, sharding) run on CPU
tensors with the oracle doing the arithmetic, and it is the checker the HIP results are
compared against.
"""
import ctypes as C
import os
import subprocess

import torch

from rectipy_amd import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "librnn_oracle.so")
_LIB = None


def build(force=False):
    src = os.path.join(HERE, "rnn_oracle.c")
    inc = os.path.join(os.path.dirname(HERE), "include")
    deps = [src, os.path.join(HERE, "rls_oracle.c"), os.path.join(HERE, "Makefile")] + [os.path.join(inc, f) for f in os.listdir(inc)]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(d) > os.path.getmtime(LIB_PATH) for d in deps)
    if stale:
        if force and os.path.exists(LIB_PATH):
            os.remove(LIB_PATH)
        subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            build()
        _LIB = _abi.declare(C.CDLL(LIB_PATH), "oracle")
    return _LIB


class OracleBackend:
    """Same methods as rectipy_amd.engine.HipBackend, host pointers, synchronous."""
    name = "oracle"

    def __init__(self):
        self.lib = load()

    def check_device(self, device):
        if torch.device(device).type != "cpu":
            raise ValueError("the oracle runs on CPU tensors only")

    def refresh_source(self, net, device):
        assert self.lib.oracle_refresh_source(C.byref(net)) == 0

    def forward(self, net, run, device):
        assert self.lib.oracle_forward(C.byref(net), C.byref(run)) == 0

    def rate_backward(self, net, bwd, device):
        assert self.lib.oracle_rate_backward(C.byref(net), C.byref(bwd)) == 0

    def qif_backward(self, net, bwd, device):
        assert self.lib.oracle_qif_backward(C.byref(net), C.byref(bwd)) == 0

    def outer_accum(self, n, T, alpha, A, lda, B, ldb, Cmat, ldc, device):
        assert self.lib.oracle_outer_accum(n, T, alpha, A, lda, B, ldb, Cmat, ldc) == 0

    def outer_accum_rect(self, m, n, T, alpha, A, lda, B, ldb, Cmat, ldc, device):
        assert self.lib.oracle_outer_accum_rect(m, n, T, alpha, A, lda, B, ldb, Cmat, ldc) == 0

    def bwd_prepare(self, net, bwd, device):
        assert self.lib.oracle_bwd_prepare(C.byref(net), C.byref(bwd)) == 0

    def bwd_partial(self, net, bwd, t, z_full, device):
        assert self.lib.oracle_bwd_partial(C.byref(net), C.byref(bwd), t, z_full) == 0

    def bwd_apply(self, net, bwd, t, z_local, device):
        assert self.lib.oracle_bwd_apply(C.byref(net), C.byref(bwd), t, z_local) == 0

    def shard_partial_rows(self, n, rows):
        return min(64, (rows + 7) // 8)  # upper bound of Q for every RC(n, rows) (chunks hold >= 8 rows, Q <= 64)

    def rls_sweep(self, rls, steps, stride, X, ldx, Y, ldy, pred, ldpred, loss, device):
        assert self.lib.oracle_rls_sweep(C.byref(rls), steps, stride, X, ldx, Y, ldy, pred, ldpred, loss) == 0

    def partial_rows(self, n):
        return min(64, (n + 7) // 8)
