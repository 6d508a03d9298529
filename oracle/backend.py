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
        _LIB = _declare(C.CDLL(LIB_PATH))
    return _LIB


def _declare(lib):
    """argtypes/restypes of the oracle's entry points (oracle/rnn_oracle.c, rls_oracle.c): the rp_* signatures of
    include/rectipy_hip.h without the stream argument, on host pointers."""
    P, fp = C.POINTER, _abi.fp
    RpNet, RpRun, RpBwd, RpRls = _abi.RpNet, _abi.RpRun, _abi.RpBwd, _abi.RpRls

    def sig(name, restype, argtypes):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes

    sig("oracle_refresh_source", C.c_int, [P(RpNet)])
    sig("oracle_forward", C.c_int, [P(RpNet), P(RpRun)])
    sig("oracle_forward_coupled", C.c_int, [P(_abi.RpMember), C.c_int32])
    sig("oracle_backward_coupled", C.c_int, [P(_abi.RpBwdMember), C.c_int32])
    sig("oracle_rate_backward", C.c_int, [P(RpNet), P(RpBwd)])
    sig("oracle_qif_backward", C.c_int, [P(RpNet), P(RpBwd)])
    sig("oracle_outer_accum", C.c_int, [C.c_int32, C.c_int32, C.c_float, fp, C.c_int64, fp, C.c_int64, fp, C.c_int64])
    sig("oracle_outer_accum_rect", C.c_int,
        [C.c_int32, C.c_int32, C.c_int32, C.c_float, fp, C.c_int64, fp, C.c_int64, fp, C.c_int64])
    sig("oracle_bwd_prepare", C.c_int, [P(RpNet), P(RpBwd)])
    sig("oracle_bwd_partial", C.c_int, [P(RpNet), P(RpBwd), C.c_int32, fp])
    sig("oracle_bwd_apply", C.c_int, [P(RpNet), P(RpBwd), C.c_int32, fp])
    sig("oracle_forward_batched_mfma", C.c_int, [P(RpNet), P(_abi.RpBatch)])
    sig("oracle_forward_batched_mfma16", C.c_int, [P(RpNet), P(_abi.RpBatch)])
    sig("oracle_rls_sweep", C.c_int, [P(RpRls), C.c_int32, C.c_int32, fp, C.c_int64, fp, C.c_int64, fp, C.c_int64, fp])
    sig("oracle_rls_mirror", C.c_int, [P(RpRls)])
    sig("oracle_edge_window", C.c_int, [P(_abi.RpEdgeMem), C.c_int32, fp, C.c_int64, fp, C.c_int64])
    sig("oracle_expf", C.c_float, [C.c_float])
    sig("oracle_tanhf", C.c_float, [C.c_float])
    sig("oracle_row_dot", C.c_float, [fp, fp, C.c_int, C.c_int, C.c_int])
    return lib


class OracleBackend:
    """Same methods as rectipy_amd.engine.HipBackend, host pointers, synchronous."""
    name = "oracle"
    # the oracle evaluates the packed-W^T adjoint (same summation order) on a dense transposed copy
    wants_dense_wt_with_packed = True

    def __init__(self):
        self.lib = load()

    def check_device(self, device):
        if torch.device(device).type != "cpu":
            raise ValueError("the oracle runs on CPU tensors only")

    def refresh_source(self, net, device):
        assert self.lib.oracle_refresh_source(C.byref(net)) == 0

    def forward(self, net, run, device):
        assert self.lib.oracle_forward(C.byref(net), C.byref(run)) == 0

    def forward_coupled(self, members, count, device):
        assert self.lib.oracle_forward_coupled(members, count) == 0

    def backward_coupled(self, members, count, device):
        assert self.lib.oracle_backward_coupled(members, count) == 0

    def rate_backward(self, net, bwd, device):
        assert self.lib.oracle_rate_backward(C.byref(net), C.byref(bwd)) == 0

    def qif_backward(self, net, bwd, device):
        assert self.lib.oracle_qif_backward(C.byref(net), C.byref(bwd)) == 0

    def outer_accum(self, n, T, alpha, A, lda, B, ldb, Cmat, ldc, device):
        assert self.lib.oracle_outer_accum(n, T, alpha, A, lda, B, ldb, Cmat, ldc) == 0

    def outer_accum_rect(self, m, n, T, alpha, A, lda, B, ldb, Cmat, ldc, device):
        assert self.lib.oracle_outer_accum_rect(m, n, T, alpha, A, lda, B, ldb, Cmat, ldc) == 0

    def outer_store_rect(self, m, n, T, alpha, A, lda, B, ldb, Cmat, ldc, device):
        C.memset(Cmat, 0, 4 * ((m - 1) * ldc + n))  # the store form has the bits of accumulating into zeros
        assert self.lib.oracle_outer_accum_rect(m, n, T, alpha, A, lda, B, ldb, Cmat, ldc) == 0

    def bwd_prepare(self, net, bwd, device):
        assert self.lib.oracle_bwd_prepare(C.byref(net), C.byref(bwd)) == 0

    def bwd_partial(self, net, bwd, t, z_full, device):
        assert self.lib.oracle_bwd_partial(C.byref(net), C.byref(bwd), t, z_full) == 0

    def bwd_apply(self, net, bwd, t, z_local, device):
        assert self.lib.oracle_bwd_apply(C.byref(net), C.byref(bwd), t, z_local) == 0

    def shard_partial_rows(self, n, rows):
        return min(64, (rows + 7) // 8)  # upper bound of Q for every RC(n, rows) (chunks hold >= 8 rows, Q <= 64)

    def forward_batched(self, net, bt, device):
        """rp_forward_batched on the oracle: the B trajectories are independent, so each is one oracle_forward on its own
        rows of the [B x ld] buffers (the batched kernel must reproduce exactly that, bit for bit)."""
        B = int(bt.batch)
        if bt.layout == _abi.RP_BATCH_MFMA:  # its own summation order: mirrored in C (oracle_forward_batched_mfma)
            assert self.lib.oracle_forward_batched_mfma(C.byref(net), C.byref(bt)) == 0
            return
        if bt.layout == _abi.RP_BATCH_MFMA16:  # the 16x16x4 form: one chain per element (oracle_forward_batched_mfma16)
            assert self.lib.oracle_forward_batched_mfma16(C.byref(net), C.byref(bt)) == 0
            return
        sampling = bool(bt.samples or bt.rec[0] or bt.rec[1])
        counts = None
        for b in range(B):
            nb = _abi.RpNet()
            C.memmove(C.byref(nb), C.byref(net), C.sizeof(_abi.RpNet))
            nb.y[_abi.RP_Y_V] = bt.v + 4 * b * bt.ldv
            nb.src[0], nb.src[1] = bt.src[0] + 4 * b * bt.ldsrc, bt.src[1] + 4 * b * bt.ldsrc
            nb.cur = bt.cur
            nb.sp_vals = None
            run = _abi.RpRun()
            run.steps, run.x_mode, run.out_var, run.sampling_steps = bt.steps, bt.x_mode, bt.out_var, 1
            if bt.x:
                run.x, run.ldx = bt.x + 4 * b * bt.ldx, B * bt.ldx
            if bt.x_mode == _abi.RP_X_PROJ:
                run.m, run.w_in, run.ldwin = bt.m, bt.w_in, bt.ldwin
            if bt.traj:
                run.traj, run.ldtraj = bt.traj + 4 * b * bt.ldtraj, B * bt.ldtraj
            if sampling:  # trajectory b's rows of the batch-major buffers
                stride = 4 * b * bt.max_samples * bt.ldsamples
                run.step0, run.cutoff, run.sampling_steps, run.max_samples = bt.step0, bt.cutoff, bt.sampling_steps, bt.max_samples
                run.acc, run.acc_count, run.sample_idx = bt.acc + 4 * b * bt.ldacc, bt.acc_count, bt.sample_idx
                run.ldsamples = bt.ldsamples
                if bt.samples:
                    run.samples = bt.samples + stride
                for q in (0, 1):
                    if bt.rec[q]:
                        run.rec[q] = bt.rec[q] + stride
            assert self.lib.oracle_forward(C.byref(nb), C.byref(run)) == 0
            counts = (run.acc_count, run.sample_idx)
        if sampling:
            bt.acc_count, bt.sample_idx = counts
        bt.cur = (bt.cur + bt.steps) % 2

    def edge_window(self, mem, steps, X, ldx, U, ldu, device):
        assert self.lib.oracle_edge_window(C.byref(mem), steps, X, ldx, U, ldu) == 0

    def rls_sweep(self, rls, steps, stride, X, ldx, Y, ldy, pred, ldpred, loss, device):
        assert self.lib.oracle_rls_sweep(C.byref(rls), steps, stride, X, ldx, Y, ldy, pred, ldpred, loss) == 0

    def rls_mirror(self, rls, device):
        assert self.lib.oracle_rls_mirror(C.byref(rls)) == 0

    def partial_rows(self, n):
        return min(64, (n + 7) // 8)
