"""Host-side driver of the C-ABI: owns the HBM-resident tensors of one recurrent population
and turns window-level requests (``steps`` Euler steps, sampling, trajectory capture, adjoint
sweep) into single calls of ``librectipy_hip.so``.

PyTorch is plumbing here (device memory, streams).  All arithmetic of the hot path happens in
the HIP kernels behind ``include/rectipy_hip.h``.

``backend`` is a test seam: the product always uses :class:`HipBackend` (and refuses CPU
tensors); tests inject an oracle-backed object with the same four methods to exercise the
host logic (sampling bookkeeping, sharding) without a GPU.
"""
import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _abi, tuning
from ._abi import RpBwd, RpNet, RpRun

MODEL_IDS = {"li_tanh": _abi.RP_MODEL_LI_TANH, "li_sigmoid": _abi.RP_MODEL_LI_SIGMOID, "qif": _abi.RP_MODEL_QIF,
             "qif_sfa": _abi.RP_MODEL_QIF_SFA, "lif": _abi.RP_MODEL_LIF, "ik": _abi.RP_MODEL_IK,
             "iku": _abi.RP_MODEL_IKU, "ik_biexp": _abi.RP_MODEL_IK_BIEXP}
SPIKING = ("qif", "qif_sfa", "lif", "ik", "iku", "ik_biexp")  # source = state variable s, threshold + reset on v
HAS_AUX = ("qif_sfa", "ik", "iku", "ik_biexp")                # second per-neuron state: adaptation x / recovery u
HAS_AUX2 = ("ik_biexp",)                                      # third state: synaptic rise variable x
NEEDS_MEANS = ("iku", "ik_biexp")                             # population means of v and spike (not shardable)

# parameter name -> slot, per model (names follow the YAML templates of the reference)
PARAM_SLOTS = {
    "li_tanh": {"tau": _abi.RP_P_TAU, "k": _abi.RP_P_K, "eta": _abi.RP_P_ETA},
    "li_sigmoid": {"tau": _abi.RP_P_TAU, "k": _abi.RP_P_K, "eta": _abi.RP_P_ETA,
                   "r_max": _abi.RP_P_RMAX, "s": _abi.RP_P_SLOPE, "v0": _abi.RP_P_V0},
    "qif": {"tau": _abi.RP_P_TAU, "k": _abi.RP_P_K, "eta": _abi.RP_P_ETA, "tau_s": _abi.RP_P_TAU_S},
    "qif_sfa": {"tau": _abi.RP_P_TAU, "k": _abi.RP_P_K, "eta": _abi.RP_P_ETA, "tau_s": _abi.RP_P_TAU_S,
                "alpha": _abi.RP_P_ALPHA, "tau_x": _abi.RP_P_TAU_X},
    "lif": {"tau": _abi.RP_P_TAU, "k": _abi.RP_P_K, "eta": _abi.RP_P_ETA, "tau_s": _abi.RP_P_TAU_S},
    "ik": {"C": _abi.RP_P_IK_C, "k": _abi.RP_P_K, "eta": _abi.RP_P_ETA, "tau_s": _abi.RP_P_TAU_S,
           "v_r": _abi.RP_P_V_R, "v_theta": _abi.RP_P_V_THETA, "g": _abi.RP_P_G, "E_r": _abi.RP_P_E_R,
           "b": _abi.RP_P_B, "tau_u": _abi.RP_P_TAU_U, "kappa": _abi.RP_P_KAPPA},
}
PARAM_SLOTS["iku"] = dict(PARAM_SLOTS["ik"])
PARAM_SLOTS["ik_biexp"] = {**{k: v for k, v in PARAM_SLOTS["ik"].items() if k != "tau_s"},
                           "tau_d": _abi.RP_P_TAU_S, "tau_r": _abi.RP_P_TAU_R}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


import contextlib

_NO_CONTEXT = contextlib.nullcontext()


class HipBackend:
    """Calls into librectipy_hip.so on the current HIP stream of the tensors' device."""
    name = "hip"

    def __init__(self, measurement: Optional[bool] = None):
        """``measurement=True``: bound to the measurement build of the library (its own handle and knobs; tests, scripts)."""
        from . import _lib
        self._lib_mod = _lib
        self.lib = _lib.load(measurement)

    def _check(self, rc):
        self._lib_mod.check(rc, self.lib)

    def check_device(self, device: torch.device):
        if device.type != "cuda":
            raise self._lib_mod.HipLibraryError(
                f"rectipy_amd integrates on MI355X only; got device '{device}'. There is no CPU fallback.")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self._check(self.lib.rp_device_check(idx))

    @staticmethod
    def _stream(device):
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    @staticmethod
    def _on(device):
        """Context that makes `device` current for the library call; a no-op object when it already is (the common
        case, and the one a per-step caller pays for)."""
        if device.index is None or device.index == torch.cuda.current_device():
            return _NO_CONTEXT
        return torch.cuda.device(device)

    def refresh_source(self, net: RpNet, device):
        with self._on(device):
            self._check(self.lib.rp_refresh_source(C.byref(net), self._stream(device)))

    def forward(self, net: RpNet, run: RpRun, device):
        with self._on(device):
            self._check(self.lib.rp_forward(C.byref(net), C.byref(run), self._stream(device)))

    def forward_coupled(self, members, count: int, device):
        with self._on(device):
            self._check(self.lib.rp_forward_coupled(members, count, self._stream(device)))

    def backward_coupled(self, members, count: int, device):
        with self._on(device):
            self._check(self.lib.rp_backward_coupled(members, count, self._stream(device)))

    def tune_forward(self, net: RpNet, device):
        """Measure the streaming configuration of the dense step kernel for this shape (a no-op for shapes that do not
        stream from HBM, and for a shape already measured or installed in this process).  WHEN this is called is the
        policy of :mod:`rectipy_amd.tuning` (on request, or for runs long enough to repay it -- not by default)."""
        with self._on(device):
            self._check(self.lib.rp_tune_forward(C.byref(net), self._stream(device)))

    def install_forward_tuning(self, net: RpNet, R: int, U: int, workgroups: int, us: float, device):
        """Install a configuration measured by an earlier process for this shape (tuning.py keeps them on disk)."""
        with self._on(device):
            self._check(self.lib.rp_install_forward_tuning(C.byref(net), R, U, workgroups, us))

    def abi_version(self) -> int:
        return int(self.lib.rp_abi_version())

    def async_status(self):
        """Raises if a resident multi-step launch (rp_resident.h: all steps of a call in one launch at n <= 1024) reported a timed-out
        wave since the last look -- the one failure such a launch can only report after the fact (rp_async_status)."""
        self._check(self.lib.rp_async_status())

    def device_name(self, device) -> str:
        """The device's name from the library (hipGetDeviceProperties), cached: torch.cuda.get_device_name would
        initialise amdsmi on its first call -- 105 ms, which was the whole "cold" cost of a process's first run."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        names = self.__dict__.setdefault("_device_names", {})
        if idx not in names:
            buf = C.create_string_buffer(256)
            self._check(self.lib.rp_device_name(idx, buf, 256))
            names[idx] = buf.value.decode("utf-8", "replace")
        return names[idx]

    def forward_tuning(self, net: RpNet) -> dict:
        R, U, G, m, us = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_float()
        self._check(self.lib.rp_forward_tuning(C.byref(net), C.byref(R), C.byref(U), C.byref(G), C.byref(us), C.byref(m)))
        return dict(rows_per_wave=R.value, chunks_in_flight=U.value, workgroups=G.value, measured=bool(m.value),
                    us_per_step=us.value)

    def rate_backward(self, net: RpNet, bwd: RpBwd, device):
        with self._on(device):
            self._check(self.lib.rp_rate_backward(C.byref(net), C.byref(bwd), self._stream(device)))

    def qif_backward(self, net: RpNet, bwd: RpBwd, device):
        with self._on(device):
            self._check(self.lib.rp_qif_backward(C.byref(net), C.byref(bwd), self._stream(device)))

    def outer_accum(self, n, T, alpha, A, lda, B, ldb, Cmat, ldc, device):
        with self._on(device):
            self._check(self.lib.rp_outer_accum(n, T, alpha, A, lda, B, ldb, Cmat, ldc, self._stream(device)))

    def outer_accum_rect(self, m, n, T, alpha, A, lda, B, ldb, Cmat, ldc, device):
        with self._on(device):
            self._check(self.lib.rp_outer_accum_rect(m, n, T, alpha, A, lda, B, ldb, Cmat, ldc, self._stream(device)))

    def outer_store_rect(self, m, n, T, alpha, A, lda, B, ldb, Cmat, ldc, device):
        with self._on(device):
            self._check(self.lib.rp_outer_store_rect(m, n, T, alpha, A, lda, B, ldb, Cmat, ldc, self._stream(device)))

    def bwd_prepare(self, net, bwd, device):
        with self._on(device):
            self._check(self.lib.rp_bwd_prepare(C.byref(net), C.byref(bwd), self._stream(device)))

    def bwd_partial(self, net, bwd, t, z_full, device):
        with self._on(device):
            self._check(self.lib.rp_bwd_partial(C.byref(net), C.byref(bwd), t, z_full, self._stream(device)))

    def bwd_apply(self, net, bwd, t, z_local, device):
        with self._on(device):
            self._check(self.lib.rp_bwd_apply(C.byref(net), C.byref(bwd), t, z_local, self._stream(device)))

    def shard_partial_rows(self, n, rows):
        return int(self.lib.rp_bwd_shard_partial_rows(n, rows))

    def forward_sharded(self, net, run, comm, device):
        with self._on(device):
            self._check(self.lib.rp_forward_sharded(C.byref(net), C.byref(run), C.byref(comm),
                                                            self._stream(device)))

    def backward_sharded(self, net, bwd, comm, device):
        with self._on(device):
            self._check(self.lib.rp_backward_sharded(C.byref(net), C.byref(bwd), C.byref(comm),
                                                                  self._stream(device)))

    def forward_batched(self, net, batch, device):
        with self._on(device):
            self._check(self.lib.rp_forward_batched(C.byref(net), C.byref(batch), self._stream(device)))

    def edge_window(self, mem, steps, X, ldx, U, ldu, device):
        with self._on(device):
            self._check(self.lib.rp_edge_window(C.byref(mem), steps, X, ldx, U, ldu, self._stream(device)))

    def rls_sweep(self, rls, steps, stride, X, ldx, Y, ldy, pred, ldpred, loss, device):
        with self._on(device):
            self._check(self.lib.rp_rls_sweep(C.byref(rls), steps, stride, X, ldx, Y, ldy, pred, ldpred, loss,
                                                      self._stream(device)))

    def rls_mirror(self, rls, device):
        with self._on(device):
            self._check(self.lib.rp_rls_mirror(C.byref(rls), self._stream(device)))

    def partial_rows(self, n):
        return int(self.lib.rp_bwd_partial_rows(n))


_DEFAULT_BACKEND = None


def default_backend():
    global _DEFAULT_BACKEND
    if _DEFAULT_BACKEND is None:
        _DEFAULT_BACKEND = HipBackend()
    return _DEFAULT_BACKEND


class Population:
    """fp32 state of one recurrent population, resident on one device.

    Parameters mirror what ``RateNet.from_pyrates`` receives (reference rectipy/nodes.py:112-164):
    a weight matrix (rows = targets), per-neuron parameter vectors, the initial state and ``dt``.
    With row sharding (``row0``/``n`` given) ``weights`` holds rows ``[row0, row0+rows)`` of the
    global ``n x n`` matrix and per-neuron vectors are local.
    """

    def __init__(self, model: str, weights, params: Dict[str, object], dt: float, device,
                 v0=None, s0=None, theta: float = 100.0, v_reset: float = -100.0,
                 n: Optional[int] = None, row0: int = 0, backend=None, aux0=None, aux2_0=None,
                 tune: Optional[bool] = None):
        if model not in MODEL_IDS:
            raise ValueError(f"unknown model '{model}'; available: {sorted(MODEL_IDS)}")
        self.model = model
        self.device = torch.device(device)
        self.backend = backend if backend is not None else default_backend()
        self.backend.check_device(self.device)
        f32 = dict(dtype=torch.float32, device=self.device)

        W = torch.as_tensor(weights) if not isinstance(weights, torch.Tensor) else weights
        if W.dim() != 2:
            raise ValueError("weights must be a 2-D matrix")
        W = W.detach().to(**f32).contiguous()
        self.rows = int(W.shape[0])
        self.n = int(W.shape[1]) if n is None else int(n)
        if W.shape[1] != self.n:
            raise RuntimeError(f"weights has {W.shape[1]} columns, expected n={self.n}")
        if n is None and W.shape[0] != W.shape[1]:
            raise RuntimeError(f"recurrent weights must be square, got {tuple(W.shape)}")
        self.row0 = int(row0)
        if self.row0 < 0 or self.row0 + self.rows > self.n:
            raise RuntimeError(f"shard rows [{self.row0}, {self.row0 + self.rows}) outside [0, {self.n})")
        self.W = W
        self.dt = float(dt)
        self.theta = float(theta)
        self.v_reset = float(v_reset)

        self.params: Dict[str, torch.Tensor] = {}
        for name in PARAM_SLOTS[model]:
            if name not in params:
                raise KeyError(f"parameter '{name}' missing for model '{model}'")
            self.params[name] = self._vec(params[name])

        self.v = self._vec(0.0 if v0 is None else v0)
        self.aux = self._vec(0.0 if aux0 is None else aux0) if model in HAS_AUX else None
        self.aux2 = self._vec(0.0 if aux2_0 is None else aux2_0) if model in HAS_AUX2 else None
        self._scratch = torch.zeros(4, **f32) if model in NEEDS_MEANS else None
        self._net_cache = None           # (key, RpNet) reused across calls, see _net()
        self.tune = tune                 # True: measure the step kernel's walk at the first step; False: never; None:
        #                                  the RECTIPY_AMD_AUTOTUNE policy (tuning.py: only runs long enough to repay it)
        self._tuned = False              # nothing left to decide: measured, installed from the cache, or ruled out
        self._cache_checked = False      # the persisted winners were looked up for this population's shape
        self.packed = None               # PackedRows of a fixed sparse W (sparse.py); the step kernel then streams it
        self._packed_version = -1        # W._version when `packed` was built (see _sync_packed)
        self.packed_t = None             # PackedRows of W^T (adjoint of a fixed sparse W), built on first use
        self.sparse_adjoint = True       # False: adjoint on the dense matrix even when a packed form exists
        self.windowed_sparse = True      # False: n > 32768 uses the plain packed form with the L2 gather
        self.sparse_block_rows = 0       # rows per block of the blocked form (0: sparse.BLOCK_ROWS)
        self.blocked_sparse = True       # False: n <= 32768 uses the per-row packed form (1.35x the bytes at p = 0.1)
        self._wt = None                  # workspace for W^T (adjoint of small networks), allocated on first use
        self.transposed_adjoint = True   # False: always use the column-strip kernels (tests, measurements)
        self.spiking = model in SPIKING
        self.src = [torch.zeros(self.n, **f32), torch.zeros(self.n, **f32)]
        self.cur = 0
        if self.spiking:
            self.set_source_local(self._vec(0.0 if s0 is None else s0))
        else:
            self.refresh()
        self.acc = torch.zeros(self.rows, **f32)
        self.acc_count = 0

    # -- helpers -------------------------------------------------------------------------
    def _vec(self, val) -> torch.Tensor:
        if isinstance(val, torch.Tensor):
            t = val.detach().to(dtype=torch.float32, device=self.device).reshape(-1)
        else:
            t = torch.as_tensor(np.asarray(val, dtype=np.float32), device=self.device).reshape(-1)
        if t.numel() == 1:
            t = t.expand(self.rows)
        if t.numel() != self.rows:
            raise RuntimeError(f"per-neuron vector has {t.numel()} entries, population has {self.rows}")
        return t.contiguous().clone()

    def _net(self) -> RpNet:
        """The population as an rp_net.  The struct is cached: the tensors it points at are updated in place for the
        lifetime of the population (only set_weights / pack_weights swap buffers), so a per-step caller does not pay
        for rebuilding it."""
        self._sync_packed()
        key = (self.W.data_ptr(), self.v.data_ptr(), id(self.packed))
        if self._net_cache is not None and self._net_cache[0] == key:
            net = self._net_cache[1]
            net.cur, net.dt, net.theta, net.v_reset = self.cur, self.dt, self.theta, self.v_reset
            return net
        net = RpNet()
        net.model = MODEL_IDS[self.model]
        net.n, net.rows, net.row0 = self.n, self.rows, self.row0
        net.ldw = self.W.stride(0)
        net.W = self.W.data_ptr()
        for name, slot in PARAM_SLOTS[self.model].items():
            net.p[slot] = self.params[name].data_ptr()
        net.y[_abi.RP_Y_V] = self.v.data_ptr()
        if self.aux is not None:
            net.y[_abi.RP_Y_AUX] = self.aux.data_ptr()
        if self.aux2 is not None:
            net.y[_abi.RP_Y_AUX2] = self.aux2.data_ptr()
        if self._scratch is not None:
            net.scratch = self._scratch.data_ptr()
        if self.packed is not None:
            net.sp_vals, net.sp_cidx = self.packed.vals.data_ptr(), self.packed.code.data_ptr()
            net.sp_rowoff = self.packed.rowoff.data_ptr()
            net.sp_block_rows = self.packed.block_rows
            if self.packed.row_ids is not None:
                net.sp_row_ids, net.sp_batch_class = self.packed.row_ids.data_ptr(), self.packed.batch_class.data_ptr()
                net.sp_phase_window = self.packed.phase_window.data_ptr()
                net.sp_phases, net.sp_batches = self.packed.phases, self.packed.batches
        net.src[0] = self.src[0].data_ptr()
        net.src[1] = self.src[1].data_ptr()
        net.cur = self.cur
        net.dt = self.dt
        net.theta = self.theta
        net.v_reset = self.v_reset
        self._net_cache = (key, net)
        return net

    # -- state ---------------------------------------------------------------------------
    @property
    def source(self) -> torch.Tensor:
        """Current presynaptic source vector (global length n): r = act(v) or s."""
        return self.src[self.cur]

    @property
    def source_local(self) -> torch.Tensor:
        return self.src[self.cur][self.row0:self.row0 + self.rows]

    def set_source_local(self, val: torch.Tensor):
        self.src[self.cur][self.row0:self.row0 + self.rows] = val

    def refresh(self):
        """Recompute the derived source r = act(v) after v was changed from the host."""
        if not self.spiking:
            self.backend.refresh_source(self._net(), self.device)

    def set_weights(self, W: torch.Tensor):
        W = W.detach().to(dtype=torch.float32, device=self.device)
        if tuple(W.shape) != (self.rows, self.n):
            raise RuntimeError(f"weights shape {tuple(W.shape)} != {(self.rows, self.n)}")
        self.W = W.contiguous()
        if self.packed is not None:
            self.pack_weights()

    def _sync_packed(self):
        """Re-pack when the dense matrix was edited in place after packing (``node["weights"]`` is the live tensor, as
        in the reference; ``set_param`` copies into it): torch bumps ``W._version`` on every in-place write."""
        if self.packed is not None and self.W._version != self._packed_version:
            self.pack_weights()

    def pack_weights(self, enable: bool = True):
        """Stream the coupling matrix in its packed sparse form from now on (forward pass only; bit-identical results,
        see sparse.py).  The dense matrix stays in place for the adjoint kernels.  Call again after changing W."""
        from . import sparse
        if not enable:
            self.packed = None
        elif self.n > sparse.WINDOW_COLUMNS and self.windowed_sparse:
            self.packed = sparse.pack_rows_windowed(self.W, row0=self.row0)  # source vector staged window by window
        elif self.n <= sparse.WINDOW_COLUMNS and self.blocked_sparse and self._block_rows() > 0:
            self.packed = sparse.pack_rows_blocked(self.W, row0=self.row0, block_rows=self._block_rows())
        else:
            self.packed = sparse.pack_rows(self.W, row0=self.row0)
        self.packed_t = None
        self._packed_version = self.W._version
        self._net_cache = None
        return self.packed

    def _block_rows(self, rows: Optional[int] = None) -> int:
        """Rows per block of the blocked packed form, 0 = per-row form.  A wave walks one block at a time, so the chip
        needs >= 16 waves x CUs blocks to stay busy: 8-row blocks from 32768 rows, 4-row blocks from 16384 rows, the
        per-row form below (measured, scripts/archive/tune_blocked.py: N = 32768, p = 0.1: 122 us vs 144 us per row-form step;
        N = 16384: 39.5 vs 42.1; N = 8192: 17.8 vs 15.2)."""
        if self.sparse_block_rows:
            return int(self.sparse_block_rows)
        rows = self.rows if rows is None else rows
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count if self.device.type == "cuda" else 256
        for rb in (8, 4):
            if rows // rb >= 16 * cus:
                return rb
        return 0

    def _packed_transpose(self):
        if self.packed_t is None:
            from . import sparse
            Wt = self.W.T.contiguous()
            if self.n > sparse.WINDOW_COLUMNS and self.windowed_sparse:
                self.packed_t = sparse.pack_rows_windowed(Wt)
            elif self.n <= sparse.WINDOW_COLUMNS and self.blocked_sparse and self._block_rows(self.n) > 0:
                self.packed_t = sparse.pack_rows_blocked(Wt, block_rows=self._block_rows(self.n))
            else:
                self.packed_t = sparse.pack_rows(Wt)
        return self.packed_t

    def reset_sampling(self):
        self.acc_count = 0  # an empty buffer: the library overwrites `acc` with the first buffered output (rp_run.acc)

    # -- forward -------------------------------------------------------------------------
    def forward(self, steps: int, x: Optional[torch.Tensor] = None, w_in: Optional[torch.Tensor] = None,
                out_var: int = _abi.RP_OUT_STATE0, step0: int = 0, cutoff: int = 0, sampling_steps: int = 1,
                samples: Optional[torch.Tensor] = None, sample_idx: int = 0, rec: Optional[dict] = None,
                traj: Optional[torch.Tensor] = None, save_v: Optional[torch.Tensor] = None,
                save_src: Optional[torch.Tensor] = None, save_acc: Optional[torch.Tensor] = None,
                spike_events: Optional[torch.Tensor] = None, spike_count: Optional[torch.Tensor] = None,
                x_target: int = _abi.RP_XT_MEMBRANE, save_aux: Optional[torch.Tensor] = None,
                save_aux2: Optional[torch.Tensor] = None, comm=None,
                save_src_full: Optional[torch.Tensor] = None, spike_src: Optional[torch.Tensor] = None,
                edges: Optional[list] = None, hist: Optional[list] = None, save_in: Optional[torch.Tensor] = None) -> int:
        """Integrate ``steps`` Euler steps in one library call.  Returns the next sample index.
        ``comm`` (an ``RpComm``): this population is a row shard and every step ends with the in-library RCCL
        all-gather of the source vector (rp_forward_sharded).

        x: None | [steps, rows] | [steps, 1] | [steps, m] together with ``w_in`` [rows, m].
        edges: dense edges into the population evaluated inside its step kernel, ``[(W_e [rows, n_e], s_e)]`` with
        ``s_e`` the edge's source vector -- ``[n_e]`` (the same vector every step) or ``[steps, n_e]``.
        """
        run, keep, sampling = self._make_run(steps, x, w_in, out_var, step0, cutoff, sampling_steps, samples, sample_idx,
                                             rec, traj, save_v, save_src, save_acc, spike_events, spike_count, x_target,
                                             save_aux, save_aux2, save_src_full, spike_src, edges, hist, save_in)
        net = self._net()
        if not self._tuned:  # static rule unless a measurement was asked for or this call is long enough to repay one
            tuning.prepare(self, net, steps)
        if comm is None:
            self.backend.forward(net, run, self.device)
        else:
            self.backend.forward_sharded(net, run, comm, self.device)
        return self._finish_run(net, run, sampling, keep)

    def _finish_run(self, net: RpNet, run: RpRun, sampling: bool, keep) -> int:
        self.cur = net.cur
        if sampling:
            self.acc_count = run.acc_count
        for e, mem in getattr(run, "_mems", ()):  # the stages of edges with memory flipped their state buffers
            mem.cur = run.edge_mem[e].cur
        del keep
        return int(run.sample_idx)

    def _make_run(self, steps, x=None, w_in=None, out_var=_abi.RP_OUT_STATE0, step0=0, cutoff=0, sampling_steps=1,
                  samples=None, sample_idx=0, rec=None, traj=None, save_v=None, save_src=None, save_acc=None,
                  spike_events=None, spike_count=None, x_target=_abi.RP_XT_MEMBRANE, save_aux=None, save_aux2=None,
                  save_src_full=None, spike_src=None, edges=None, hist=None, save_in=None):
        """The rp_run of one call (see :meth:`forward`): (run, tensors to keep alive until the call returned, sampling?).
        Edge sources given as ``None`` are left for rp_forward_coupled to fill in."""
        run = RpRun()
        run.steps = int(steps)
        run.step0 = int(step0)
        keep = []  # keep converted tensors alive until the call returned
        if x is None:
            run.x_mode = _abi.RP_X_NONE
        else:
            if x.dim() != 2 or x.shape[0] < steps:
                raise RuntimeError(f"inputs must be [steps, m] with at least {steps} rows, got {tuple(x.shape)}")
            x = x.detach().to(dtype=torch.float32, device=self.device)
            if x.stride(1) != 1:
                x = x.contiguous()
            keep.append(x)
            run.x = x.data_ptr()
            run.ldx = x.stride(0) if x.shape[0] > 1 else x.shape[1]
            if w_in is not None:
                w_in = w_in.detach().to(dtype=torch.float32, device=self.device).contiguous()
                if tuple(w_in.shape) != (self.rows, x.shape[1]):
                    raise RuntimeError(f"input weights {tuple(w_in.shape)} do not map {x.shape[1]} inputs to {self.rows} neurons")
                keep.append(w_in)
                run.x_mode, run.m = _abi.RP_X_PROJ, x.shape[1]
                run.w_in, run.ldwin = w_in.data_ptr(), w_in.stride(0)
            elif x.shape[1] == self.rows:
                run.x_mode = _abi.RP_X_FULL
            elif x.shape[1] == 1:
                run.x_mode = _abi.RP_X_BCAST
            else:
                raise RuntimeError(f"input of width {x.shape[1]} does not match the {self.rows} neurons of the population")
        run.out_var = out_var
        run.x_target = int(x_target)
        run.cutoff = int(cutoff)
        run.sampling_steps = int(sampling_steps)

        def rowbuf(t, name, min_rows):
            if t is None:
                return None, 0
            if t.dtype != torch.float32 or t.device != self.device or t.dim() != 2 or t.stride(1) != 1 \
                    or t.shape[1] < self.rows or t.shape[0] < min_rows:
                raise RuntimeError(f"{name} must be a float32 [>= {min_rows}, >= {self.rows}] row-major tensor on {self.device}")
            return t.data_ptr(), t.stride(0)

        run.traj, run.ldtraj = rowbuf(traj, "traj", steps)
        ld_save = 0
        for t, field in ((save_v, 0), (save_aux, 1), (save_aux2, 2), (save_src, "save_src"), (save_acc, "save_acc"),
                         (save_in, "save_in")):
            p, ld = rowbuf(t, "save buffer", steps)
            if p is not None:
                if ld_save and ld != ld_save:
                    raise RuntimeError("all save buffers must share one leading dimension")
                ld_save = ld
                if isinstance(field, int):
                    run.save_state[field] = p
                else:
                    setattr(run, field, p)
        run.ldsave = ld_save

        sampling = samples is not None or bool(rec)
        if sampling:
            run.acc = self.acc.data_ptr()
            run.acc_count = self.acc_count
            run.sample_idx = int(sample_idx)
            ld_s, max_rows = 0, None
            bufs = [("samples", samples)] + [(f"rec[{k}]", v) for k, v in (rec or {}).items()]
            for name, t in bufs:
                if t is None:
                    continue
                p, ld = rowbuf(t, name, 0)
                if ld_s and ld != ld_s:
                    raise RuntimeError("sample/record buffers must share one leading dimension")
                ld_s = ld
                max_rows = t.shape[0] if max_rows is None else min(max_rows, t.shape[0])
            run.samples = _ptr(samples)
            for k, t in (rec or {}).items():
                run.rec[int(k)] = _ptr(t)
            run.ldsamples = ld_s
            run.max_samples = max_rows or 0
        if save_src_full is not None:
            t = save_src_full
            if t.dtype != torch.float32 or t.device != self.device or t.dim() != 2 or t.stride(1) != 1 \
                    or t.shape[1] < self.n or t.shape[0] < steps:
                raise RuntimeError(f"save_src_full must be a float32 [>= {steps}, >= {self.n}] row-major tensor")
            run.save_src_full, run.ldfull = t.data_ptr(), t.stride(0) if t.shape[0] > 1 else t.shape[1]
        if spike_src is not None:  # first step detects spikes on these values (rp_run.spike_src)
            if spike_src.dtype != torch.float32 or spike_src.device != self.device or spike_src.numel() != self.rows \
                    or not spike_src.is_contiguous():
                raise RuntimeError(f"spike_src must be a contiguous float32 [{self.rows}] tensor on {self.device}")
            run.spike_src = spike_src.data_ptr()
            keep.append(spike_src)
        if spike_count is not None:
            run.spike_events = spike_events.data_ptr()
            run.spike_capacity = spike_events.shape[0]
            run.spike_count = spike_count.data_ptr()

        if edges:
            if len(edges) > _abi.RP_MAX_EDGES:
                raise RuntimeError(f"at most {_abi.RP_MAX_EDGES} dense edges per population, got {len(edges)}")
            run.n_edges = len(edges)
            for e, edge in enumerate(edges):
                We, se = edge[0], edge[1]
                pk = edge[2] if len(edge) > 2 else None   # the edge matrix in packed form (sparse.PackedRows), coupled calls only
                if pk is not None:
                    sp = run.edge_sp[e]
                    sp.vals, sp.cidx, sp.rowoff, sp.block_rows = pk.vals.data_ptr(), pk.code.data_ptr(), pk.rowoff.data_ptr(), pk.block_rows
                    if pk.row_ids is not None:
                        sp.row_ids, sp.batch_class = pk.row_ids.data_ptr(), pk.batch_class.data_ptr()
                        sp.phase_window, sp.phases, sp.batches = pk.phase_window.data_ptr(), pk.phases, pk.batches
                    keep.append(pk)
                    if getattr(self, "_edge_sum", None) is None:
                        self._edge_sum = torch.zeros(self.rows, dtype=torch.float32, device=self.device)
                    run.edge_sum = self._edge_sum.data_ptr()
                mem = edge[3] if len(edge) > 3 else None  # the edge's memory (EdgeMemory): delay buffer / recursive filter
                if We is None:  # the identity edge: E_e = s_e
                    run.edge_w[e], run.edge_ld[e], run.edge_n[e] = None, self.rows, self.rows
                    n_e = self.rows
                else:
                    if We.dtype != torch.float32 or We.device != self.device or We.dim() != 2 or We.shape[0] != self.rows \
                            or We.stride(1) != 1:
                        raise RuntimeError(f"edge {e}: weights must be a float32 [{self.rows}, n_e] row-major tensor on {self.device}")
                    keep.append(We)
                    run.edge_w[e], run.edge_ld[e], run.edge_n[e] = We.data_ptr(), We.stride(0), We.shape[1]
                    n_e = We.shape[1]
                if mem is not None:
                    if mem.n != n_e:
                        raise RuntimeError(f"edge {e}: its memory holds {mem.n} channels, the edge has {n_e} columns")
                    mem.fill(run.edge_mem[e])
                    keep.append(mem)
                    run._mems = getattr(run, "_mems", []) + [(e, mem)]
                if se is None:
                    continue
                se = se.detach()
                if se.dtype != torch.float32 or se.device != self.device or se.shape[-1] != n_e or se.stride(-1) != 1 \
                        or se.dim() not in (1, 2) or (se.dim() == 2 and se.shape[0] < steps):
                    raise RuntimeError(f"edge {e}: the source must be a float32 [{n_e}] or [steps, {n_e}] tensor")
                keep.append(se)
                run.edge_src[e] = se.data_ptr()
                run.edge_src_ld[e] = 0 if se.dim() == 1 else (se.stride(0) if se.shape[0] > 1 else se.shape[1])
        if hist is not None:
            for q in (0, 1, 2):
                h = hist[q]
                if h.dtype != torch.float32 or h.device != self.device or h.numel() != self.rows or not h.is_contiguous():
                    raise RuntimeError(f"hist[{q}] must be a contiguous float32 [{self.rows}] tensor on {self.device}")
                run.hist[q] = h.data_ptr()
            keep.extend(hist)
        return run, keep, sampling

    # -- adjoint -------------------------------------------------------------------------
    def backward(self, traj_v, traj_r, grad_out, grad_var, lam, u_traj, traj_acc=None,
                 grad_tau=None, grad_k=None, grad_eta=None, grad_x=None, lam_s=None, grad_tau_s=None,
                 slope: float = 0.5, x: Optional[torch.Tensor] = None, x_target: int = _abi.RP_XT_MEMBRANE,
                 traj_aux=None, lam_x=None, grad_params: Optional[Dict[str, torch.Tensor]] = None,
                 traj_aux2=None, lam_x2=None, exchange=None, _args_only: bool = False):
        """Adjoint sweep over a stored window (rp_rate_backward; rp_qif_backward for the spiking models, which
        additionally take the adjoint of s, the surrogate slope and the window's input for grad_tau).

        Row shard (``rows < n``): every per-neuron tensor is local and ``exchange`` provides the cross-rank sum of the
        full-length partial W_p^T u_p -- ``exchange.comm`` (an ``RpComm``: the whole sweep is one library call,
        rp_backward_sharded) or ``exchange.reduce_scatter(z_full, z_local)`` called once per step."""
        T = int(traj_v.shape[0])
        local = self.rows
        sharded = self.rows != self.n or _args_only == "stepwise"  # (the step-wise entry points: no W^T workspace)
        self._sync_packed()
        b = RpBwd()
        b.steps = T
        b.traj_v, b.traj_r, b.traj_acc = traj_v.data_ptr(), traj_r.data_ptr(), _ptr(traj_acc)
        b.ld = traj_v.stride(0) if T > 1 else max(traj_v.shape[1], local)
        b.grad_var = grad_var
        b.grad_out = _ptr(grad_out)
        b.ldg = 0 if grad_out is None else (grad_out.stride(0) if T > 1 else grad_out.shape[1])
        b.lam = lam.data_ptr()
        b.u_traj = u_traj.data_ptr()
        q = self.backend.shard_partial_rows(self.n, local) if sharded else self.backend.partial_rows(self.n)
        partials = torch.empty(q, self.n, dtype=torch.float32, device=self.device)
        b.partials = partials.data_ptr()
        if (self.rows == self.n) and self.packed is not None and self.sparse_adjoint:  # (also for the step-wise entry points)
            # fixed sparse W: the sweep streams the packed rows of W^T (row-dot order of the transposed-copy sweep)
            pt = self._packed_transpose()
            if pt is not None:
                b.spt_vals, b.spt_cidx, b.spt_rowoff = pt.vals.data_ptr(), pt.code.data_ptr(), pt.rowoff.data_ptr()
                b.spt_block_rows = pt.block_rows
                if pt.row_ids is not None:
                    b.spt_row_ids, b.spt_batch_class = pt.row_ids.data_ptr(), pt.batch_class.data_ptr()
                    b.spt_phase_window, b.spt_phases, b.spt_batches = pt.phase_window.data_ptr(), pt.phases, pt.batches
        # (a backend may ask for the W^T workspace next to the packed rows of W^T: `wants_dense_wt_with_packed`)
        if not sharded and ((self.n <= _abi.RP_BWD_TRANSPOSE_MAX_N and self.transposed_adjoint and not b.spt_vals)
                            or (b.spt_vals and getattr(self.backend, "wants_dense_wt_with_packed", False))):
            # small networks: the sweep keeps W^T in this workspace and runs one fused launch per step (rp_bwd.wt)
            if self._wt is None:
                self._wt = torch.empty(self.n, (self.n + 3) // 4 * 4, dtype=torch.float32, device=self.device)
            b.wt, b.ldwt = self._wt.data_ptr(), self._wt.stride(0)
        b.grad_tau, b.grad_k, b.grad_eta = _ptr(grad_tau), _ptr(grad_k), _ptr(grad_eta)
        b.grad_x = _ptr(grad_x)
        b.ldgx = 0 if grad_x is None else (grad_x.stride(0) if T > 1 else grad_x.shape[1])
        for name, t in (grad_params or {}).items():  # model-specific parameters by engine name (sigmoid, spiking)
            b.grad_p[PARAM_SLOTS[self.model][name]] = t.data_ptr()
        keep = [partials]  # returned so the caller keeps the workspaces alive until the stream has consumed them
        if self.spiking:
            if self.model in NEEDS_MEANS:
                if self.rows != self.n:
                    raise NotImplementedError(f"the '{self.model}' model (population-mean coupling) cannot be row-sharded")
                means = torch.zeros(4, dtype=torch.float32, device=self.device)
                keep.append(means)
                b.means = means.data_ptr()
                b.traj_aux2, b.lam4 = _ptr(traj_aux2), _ptr(lam_x2)
            b.lam2 = lam_s.data_ptr()
            b.grad_tau_s = _ptr(grad_tau_s)
            b.slope = float(slope)
            b.x_target = int(x_target)
            b.traj_aux, b.lam3 = _ptr(traj_aux), _ptr(lam_x)
            if x is None:
                b.x_mode = _abi.RP_X_NONE
            else:
                x = x.detach().to(dtype=torch.float32, device=self.device).contiguous()
                if x.shape[1] not in (1, local) or x.shape[0] < T:
                    raise RuntimeError(f"window input must be [{T}, 1] or [{T}, {local}], got {tuple(x.shape)}")
                b.x_mode = _abi.RP_X_FULL if x.shape[1] == local and local != 1 else _abi.RP_X_BCAST
                b.x, b.ldx = x.data_ptr(), x.shape[1]
                keep.append(x)
        net = self._net()
        if _args_only:
            return net, b, keep
        sharded = self.rows != self.n
        if not sharded:
            (self.backend.qif_backward if self.spiking else self.backend.rate_backward)(net, b, self.device)
            return keep
        if exchange is None:
            raise RuntimeError("the adjoint of a row shard needs an `exchange` (cross-rank sum of W_p^T u_p)")
        f32 = dict(dtype=torch.float32, device=self.device)
        if getattr(exchange, "comm", None) is not None:  # whole sweep in one call, RCCL reduce-scatter in-library
            work = torch.empty(self.n + local, **f32)
            keep.append(work)
            exchange.comm.work = work.data_ptr()
            self.backend.backward_sharded(net, b, exchange.comm, self.device)
            exchange.comm.work = None
            return keep
        z_full, z_local = torch.empty(self.n, **f32), torch.empty(local, **f32)
        self.backend.bwd_prepare(net, b, self.device)
        for t in range(T - 1, -1, -1):
            self.backend.bwd_partial(net, b, t, z_full.data_ptr(), self.device)
            exchange.reduce_scatter(z_full, z_local)
            self.backend.bwd_apply(net, b, t, z_local.data_ptr(), self.device)
        return keep

    def _backward_args(self, **kw):
        """(rp_net, rp_bwd, tensors to keep alive) of an adjoint sweep -- the arguments :meth:`backward` would hand to the
        library -- for callers that drive the step-wise entry points themselves (sharded.simulate_sharded_backward)."""
        return self.backward(_args_only=kw.pop("_args_only", True), **kw)

    def outer_accum(self, alpha: float, A: torch.Tensor, B: torch.Tensor, Cmat: torch.Tensor):
        """Cmat += alpha * A^T @ B on the matrix cores (A, B: [T, n])."""
        T = int(A.shape[0])
        self.backend.outer_accum(self.n, T, float(alpha), A.data_ptr(), A.stride(0) if T > 1 else A.shape[1],
                                 B.data_ptr(), B.stride(0) if T > 1 else B.shape[1], Cmat.data_ptr(),
                                 Cmat.stride(0), self.device)


    def outer_product(self, alpha: float, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
        """alpha * A^T @ B as a NEW [n, n] matrix (A, B: [T, n]): the GEMM's store form -- no zero-fill pass over the
        result and no read of it in the epilogue; the bits of ``outer_accum`` into zeros."""
        T = int(A.shape[0])
        out = torch.empty(self.n, self.n, dtype=torch.float32, device=self.device)
        self.backend.outer_store_rect(self.n, self.n, T, float(alpha), A.data_ptr(), A.stride(0) if T > 1 else A.shape[1],
                                      B.data_ptr(), B.stride(0) if T > 1 else B.shape[1], out.data_ptr(), out.stride(0),
                                      self.device)
        return out


class InstantPopulation(Population):
    """A memory-less activation node BETWEEN populations (reference ``InstantNode``, rectipy/nodes.py:14-51) as a member
    of a coupled call (RP_MODEL_INSTANT): no state, no recurrent matrix -- its step is ``out = act(I)`` with I the node
    input of the step (edge sums, plus an external input if it has one).  With ``act="identity"`` and identity edges it
    also carries the partial sum of an edge list longer than RP_MAX_EDGES.  ``source`` is the node's last output."""

    ACTS = {"identity": _abi.RP_ACT_IDENTITY, "tanh": _abi.RP_ACT_TANH, "sigmoid": _abi.RP_ACT_SIGMOID}

    def __init__(self, n: int, act: str, device, backend=None):
        if act not in self.ACTS:
            raise ValueError(f"activation '{act}' has no in-kernel form; available: {sorted(self.ACTS)}")
        self.model, self.act = "instant", act
        self.device = torch.device(device)
        self.backend = backend if backend is not None else default_backend()
        self.backend.check_device(self.device)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.rows = self.n = int(n)
        self.row0 = 0
        self.W = self.v = self.aux = self.aux2 = self.packed = self.packed_t = None
        self.params = {}
        self.dt, self.theta, self.v_reset = 1.0, 0.0, 0.0
        self.spiking = False
        self.src = [torch.zeros(self.n, **f32), torch.zeros(self.n, **f32)]
        self.cur = 0
        self.acc = torch.zeros(self.rows, **f32)
        self.acc_count = 0
        self.tune, self._tuned = False, True
        self._net_struct = None

    def _sync_packed(self):
        pass

    def refresh(self):
        pass

    def _net(self) -> RpNet:
        if self._net_struct is None:
            net = RpNet()
            net.model, net.act = _abi.RP_MODEL_INSTANT, self.ACTS[self.act]
            net.n = net.rows = self.n
            net.ldw = self.n
            net.src[0], net.src[1] = self.src[0].data_ptr(), self.src[1].data_ptr()
            net.dt = 1.0
            self._net_struct = net
        self._net_struct.cur = self.cur
        return self._net_struct


class EdgeMemory:
    """Device state of the MEMORY of a stateful edge -- the delay buffer of ``LinearMemory``, the recursive filter of
    ``LinearFilter``, both for ``LinearMemoryFilter`` (reference rectipy/edges.py:68-147) -- as an ``rp_edge_mem``: the stage
    in front of the edge's projection, advanced by the library once per step (rp_run.edge_mem inside rp_forward /
    rp_forward_coupled; :meth:`window` = rp_edge_window for an acyclic edge, whose source window is known in advance).

    ``state``: DELAY [depth] (the buffer's common row), FILTER [n], DELAY_FILTER [depth, n] (row c = column c of the
    reference's buffer)."""

    KINDS = {"delay": _abi.RP_EMEM_DELAY, "filter": _abi.RP_EMEM_FILTER, "delay_filter": _abi.RP_EMEM_DELAY_FILTER}

    def __init__(self, kind: str, n: int, device, backend=None, F: Optional[torch.Tensor] = None, depth: int = 0,
                 col_chan=None):
        self.kind, self.n, self.depth = kind, int(n), int(depth)
        self.device = torch.device(device)
        self.backend = backend if backend is not None else default_backend()
        self.backend.check_device(self.device)
        f32 = dict(dtype=torch.float32, device=self.device)
        shape = {"delay": (self.depth,), "filter": (self.n,), "delay_filter": (self.depth, self.n)}[kind]
        self.state = [torch.zeros(shape, **f32), torch.zeros(shape, **f32)]
        self.cur = 0
        self.u = torch.zeros(self.n, **f32)
        self.save, self.save_row0 = None, 0   # optional [steps, ld] window of the stage's outputs u_t (rp_edge_mem.save_u) and the
        #                                       row the next library call starts at
        self.F = None
        if kind != "delay":
            self.set_filter(F)
        self.col_chan = self.col_chan_host = None
        if kind != "filter":
            cc = np.ascontiguousarray(np.asarray(col_chan, dtype=np.int32).reshape(-1))
            if cc.shape[0] != self.depth:
                raise RuntimeError(f"col_chan has {cc.shape[0]} entries, the delay buffer {self.depth} columns")
            self.col_chan_host = cc
            self.col_chan = torch.as_tensor(cc, device=self.device)

    def set_filter(self, F: torch.Tensor):
        F = F.detach().to(dtype=torch.float32, device=self.device)
        if tuple(F.shape) != (self.n, self.n):
            raise RuntimeError(f"filter weights {tuple(F.shape)} do not match the edge's {self.n} inputs")
        self.F = F if F.stride(1) == 1 else F.contiguous()

    @property
    def current(self) -> torch.Tensor:
        return self.state[self.cur]

    def load(self, state: torch.Tensor):
        self.state[self.cur].copy_(state.detach().to(dtype=torch.float32, device=self.device).reshape(self.state[0].shape))

    def fill(self, m: "_abi.RpEdgeMem"):
        m.kind, m.n, m.depth = self.KINDS[self.kind], self.n, self.depth
        if self.col_chan is not None:
            m.col_chan, m.col_chan_host = self.col_chan.data_ptr(), self.col_chan_host.ctypes.data
        if self.F is not None:
            m.F, m.ldf = self.F.data_ptr(), self.F.stride(0)
        m.state[0], m.state[1] = self.state[0].data_ptr(), self.state[1].data_ptr()
        m.cur = self.cur
        m.u = self.u.data_ptr()
        if self.save is not None:
            m.save_u, m.ld_save = self.save[self.save_row0:].data_ptr(), self.save.stride(0)

    def window(self, X: torch.Tensor) -> torch.Tensor:
        """U [T, n]: the stage's output for the T consecutive inputs X[t]; the state moves on by T steps."""
        T = int(X.shape[0])
        X = X.detach().to(dtype=torch.float32, device=self.device)
        if X.dim() != 2 or X.shape[1] != self.n:
            raise RuntimeError(f"the edge's memory takes [T, {self.n}] inputs, got {tuple(X.shape)}")
        if X.stride(1) != 1:
            X = X.contiguous()
        ld = (self.n + 3) // 4 * 4   # (16-byte aligned rows: a wide filter reads row t - 1 as its source vector)
        U = torch.empty(T, ld, dtype=torch.float32, device=self.device)
        m = _abi.RpEdgeMem()
        self.fill(m)
        self.backend.edge_window(m, T, X.data_ptr(), X.stride(0) if T > 1 else self.n, U.data_ptr(), ld, self.device)
        self.cur = m.cur
        return U[:, :self.n]


def forward_coupled(members: list) -> list:
    """One library call for a whole window of SEVERAL populations joined by dense edges (rp_forward_coupled): every
    Euler step steps the populations in list order, one launch each, the edge products inside the step kernels.

    ``members``: dicts with ``pop`` (:class:`Population`, or :class:`InstantPopulation` for an activation node between
    populations), ``run`` (keyword arguments of :meth:`Population.forward` without ``edges`` / ``comm``) and ``edges``:
    ``[(W_e [rows, n_source] | None = identity, source member index, RP_EDGE_* mode)]``, optionally with the edge matrix's
    packed form (``sparse.PackedRows``) as a fourth entry -- a member whose W is packed and whose edges all carry one streams
    only the stored nonzeros (rp_run.edge_sp; same bits) -- and the edge's memory (:class:`EdgeMemory`) as a fifth.
    Returns the next sample index of every member."""
    members, real = _chain_long_edge_lists(members)
    count = len(members)
    arr = (_abi.RpMember * count)()
    built = []
    for m, spec in enumerate(members):
        pop: Population = spec["pop"]
        edges = spec.get("edges") or []
        run, keep, sampling = pop._make_run(edges=[(edge[0], None, edge[3] if len(edge) > 3 else None, edge[4] if len(edge) > 4 else None)
                                                   for edge in edges], **spec["run"])
        net = pop._net()
        for e, edge in enumerate(edges):
            arr[m].edge_source[e], arr[m].edge_mode[e] = int(edge[1]), int(edge[2])
        arr[m].net, arr[m].run = C.pointer(net), C.pointer(run)
        built.append((pop, net, run, sampling, keep))
    first = members[0]["pop"]
    if any(spec["pop"].device != first.device for spec in members):
        raise RuntimeError("the members of a coupled call must live on one device: " +
                           ", ".join(sorted({str(spec["pop"].device) for spec in members})))
    if count > _abi.RP_MAX_MEMBERS:
        raise RuntimeError(f"at most {_abi.RP_MAX_MEMBERS} members per coupled call, got {count}")
    first.backend.forward_coupled(arr, count, first.device)
    done = [pop._finish_run(net, run, sampling, keep) for pop, net, run, sampling, keep in built]
    return [done[i] for i in real]


def _chain_long_edge_lists(members: list):
    """A member with more than RP_MAX_EDGES edges gets partial-sum members in front of it: identity activation nodes of its
    width (InstantPopulation, cached on the population) that add the first RP_MAX_EDGES edges, then -- through an identity edge
    from the previous partial sum -- the next RP_MAX_EDGES - 1, and so on; the member itself keeps an identity edge from the
    last partial sum and the remaining edges.  E = ((E_0 + E_1) + ...) is continued from member to member in the same order,
    so the node input has the bits of one long edge list.  Returns (members, positions of the caller's members)."""
    K = _abi.RP_MAX_EDGES
    if all(len(spec.get("edges") or []) <= K for spec in members):
        return members, list(range(len(members)))
    new_index, out = {}, []
    for m, spec in enumerate(members):  # positions first: edges name their sources by index
        edges = spec.get("edges") or []
        extra = 0
        left = len(edges)
        if left > K:
            extra, left = 1, left - K
            while left + 1 > K:
                extra, left = extra + 1, left - (K - 1)
        new_index[m] = len(out) + extra
        out.extend([None] * (extra + 1))
    for m, spec in enumerate(members):
        pop = spec["pop"]
        edges = [(e[0], new_index[e[1]]) + tuple(e[2:]) for e in (spec.get("edges") or [])]
        at = new_index[m]
        n_sums = at - (new_index[m - 1] + 1 if m > 0 else 0)
        if n_sums == 0:
            out[at] = dict(spec, edges=edges)
            continue
        sums = getattr(pop, "_edge_sums", None)
        if sums is None or len(sums) < n_sums:
            sums = pop._edge_sums = [InstantPopulation(pop.rows, "identity", pop.device, backend=pop.backend) for _ in range(n_sums)]
        run0 = spec["run"]
        first = at - n_sums
        taken = 0
        for k in range(n_sums):
            part = edges[taken:taken + (K if k == 0 else K - 1)]
            taken += len(part)
            if k > 0:
                part = [(None, first + k - 1, _abi.RP_EDGE_OUT)] + part
            out[first + k] = dict(pop=sums[k], run=dict(steps=run0["steps"], step0=run0.get("step0", 0), out_var=_abi.RP_OUT_SOURCE),
                                  edges=part)
        out[at] = dict(spec, edges=[(None, at - 1, _abi.RP_EDGE_OUT)] + edges[taken:])
    return out, [new_index[m] for m in range(len(members))]


def backward_coupled(members: list) -> list:
    """The adjoint of a :func:`forward_coupled` window in one library call (rp_backward_coupled): every step of the
    backward sweep first carries the input gradient of every edge's target through W_e^T into the output gradient of its
    source, then steps every member's own adjoint.

    ``members`` (the order of the forward call): dicts with ``pop``, ``bwd`` (keyword arguments of
    :meth:`Population.backward`; ``grad_out`` [T, rows] is required -- and written -- for every member an edge reads,
    ``grad_x`` for every member with edges; for an activation-node member -- an :class:`InstantPopulation` --
    ``dict(instant=True, traj_r=its outputs [T, n], grad_out=..., grad_x=...)``), ``edges`` and, for a member whose output an
    RP_EDGE_LIVE edge of a LATER member reads (a post-update read), ``final_src``: its source vector at the end of the window.
    An edge is ``(W_e, source index, RP_EDGE_* mode[, packed transposed form])`` or a dict ``W / source / mode / packed_t /
    memory ("delay" | "filter" | None) / edge`` -- ``edge``: the front end's edge object, which names the filter matrix or
    the delays of an edge with memory (the gradient then passes through the memory: rp_bwd_member.edge_mem_kind).
    Returns the tensors to keep alive until the stream has run the sweep."""
    count = len(members)
    if count > _abi.RP_MAX_MEMBERS:
        raise RuntimeError(f"at most {_abi.RP_MAX_MEMBERS} members per coupled call, got {count}")
    arr = (_abi.RpBwdMember * count)()
    keep_all = []
    norm = []   # every edge as a dict
    for spec in members:
        row = []
        for edge in spec.get("edges") or []:
            if isinstance(edge, dict):
                row.append(dict(W=edge["W"], source=edge["source"], mode=edge["mode"], packed_t=edge.get("packed_t"),
                                memory=edge.get("memory"), edge=edge.get("edge")))
            else:
                row.append(dict(W=edge[0], source=edge[1], mode=edge[2], packed_t=edge[3] if len(edge) > 3 else None, memory=None, edge=None))
        norm.append(row)
    work_floats = [0] * count
    for m, spec in enumerate(members):
        pop: Population = spec["pop"]
        if len(norm[m]) > _abi.RP_MAX_EDGES:
            raise RuntimeError(f"member {m}: at most {_abi.RP_MAX_EDGES} edges per member in the coupled adjoint, got {len(norm[m])}")
        for edge in norm[m]:
            We, src = edge["W"], members[edge["source"]]["pop"]
            if We.dim() != 2 or We.shape != (pop.rows, src.n) or We.stride(1) != 1 or We.dtype != torch.float32 \
                    or We.device != pop.device:
                raise RuntimeError(f"member {m}: an edge matrix must be a float32 [{pop.rows}, {src.n}] matrix with unit "
                                   f"column stride on {pop.device}")
            work_floats[m] += pop.backend.shard_partial_rows(src.n, pop.rows) * src.n  # (every edge its own slice)
    first: Population = members[0]["pop"]
    for m, spec in enumerate(members):
        pop = spec["pop"]
        work = torch.empty(max(work_floats[m], 4), dtype=torch.float32, device=pop.device)
        extra = []
        if spec["bwd"].get("instant"):   # an activation node: its outputs, its output gradient (in / out), its input gradient (out)
            kw = spec["bwd"]
            net, b, keep, z = pop._net(), RpBwd(), [], None
            T = int(kw["traj_r"].shape[0])
            b.steps = T
            b.traj_r, b.ld = kw["traj_r"].data_ptr(), kw["traj_r"].stride(0) if T > 1 else pop.rows
            b.grad_out, b.ldg = kw["grad_out"].data_ptr(), kw["grad_out"].stride(0) if T > 1 else pop.rows
            b.grad_x, b.ldgx = kw["grad_x"].data_ptr(), kw["grad_x"].stride(0) if T > 1 else pop.rows
            b.grad_var = _abi.RP_OUT_SOURCE
        else:
            net, b, keep = pop._backward_args(_args_only="stepwise", **spec["bwd"])
            z = torch.empty(pop.n, dtype=torch.float32, device=pop.device)
            arr[m].z = z.data_ptr()
        arr[m].net, arr[m].bwd = C.addressof(net), C.addressof(b)
        arr[m].n_edges = len(norm[m])
        for e, edge in enumerate(norm[m]):
            We = edge["W"]
            arr[m].edge_w[e], arr[m].edge_ld[e] = We.data_ptr(), We.stride(0)
            arr[m].edge_source[e], arr[m].edge_mode[e] = int(edge["source"]), int(edge["mode"])
            pk = edge["packed_t"]   # the TRANSPOSED edge matrix in packed form (a fixed sparse edge)
            if pk is not None:
                sp = arr[m].edge_spt[e]
                sp.vals, sp.cidx, sp.rowoff, sp.block_rows = pk.vals.data_ptr(), pk.code.data_ptr(), pk.rowoff.data_ptr(), pk.block_rows
                if pk.row_ids is not None:
                    sp.row_ids, sp.batch_class = pk.row_ids.data_ptr(), pk.batch_class.data_ptr()
                    sp.phase_window, sp.phases, sp.batches = pk.phase_window.data_ptr(), pk.phases, pk.batches
            if edge["memory"] == "filter":   # ly(t) = W_e^T gx(t) + F^T ly(t + 1) goes into the source's output gradient
                n_s = We.shape[1]
                F = edge["edge"].filter.detach().to(dtype=torch.float32, device=pop.device)
                F = F if F.stride(1) == 1 else F.contiguous()
                lam = torch.zeros(n_s, dtype=torch.float32, device=pop.device)
                fwork = torch.empty(max(4, pop.backend.shard_partial_rows(n_s, n_s) * n_s), dtype=torch.float32, device=pop.device)
                arr[m].edge_mem_kind[e] = _abi.RP_EMEM_FILTER
                arr[m].edge_F[e], arr[m].edge_ldf[e] = F.data_ptr(), F.stride(0)
                arr[m].edge_lam[e], arr[m].edge_fwork[e] = lam.data_ptr(), fwork.data_ptr()
                extra += [F, lam, fwork]
            elif edge["memory"] == "delay":  # u_t = x_{t - d}[j] * ones: d the smallest delay, j the channel written there
                obj = edge["edge"]
                arr[m].edge_mem_kind[e] = _abi.RP_EMEM_DELAY
                arr[m].edge_delay[e], arr[m].edge_chan[e] = int(obj._cols[0]), int(obj._src[0])
            elif edge["memory"] is not None:
                raise RuntimeError(f"member {m}: no adjoint through an edge memory of kind '{edge['memory']}'")
        arr[m].edge_work = work.data_ptr()
        final_src = spec.get("final_src")  # r at the end of the window: a later member's RP_EDGE_LIVE edge read it (a post-update read)
        if final_src is not None:
            if final_src.dtype != torch.float32 or final_src.device != pop.device or final_src.numel() != pop.n or not final_src.is_contiguous():
                raise RuntimeError(f"member {m}: final_src must be a contiguous float32 [{pop.n}] tensor on {pop.device}")
            arr[m].final_src = final_src.data_ptr()
        keep_all.append((net, b, keep, z, work, final_src, [edge["W"] for edge in norm[m]], [edge["packed_t"] for edge in norm[m]], extra))
    first.backend.backward_coupled(arr, count, first.device)
    return keep_all


class BatchedPopulation:
    """B independent trajectories of ONE rate population (same W, same per-neuron parameters; different states and
    inputs) advanced together: every Euler step streams W once for all of them (``rp_forward_batched``).  The reference
    has no batch axis -- B trials of a network are B passes of its time loop, each streaming W again.

    B in {2, 4, 8}: vector ALUs, state batch-major ([B, n]); trajectory b is BIT-IDENTICAL to an unbatched
    :class:`Population` run on the same state.  9 <= B <= 32 (or ``mfma=True``): the contraction runs on the matrix cores
    (exact-fp32 MFMA) with up to 32 trajectories per pass over W; equal to the unbatched run up to fp32 rounding (its own
    summation order, mirrored by the oracle).  Two matrix-core forms: "mfma16" (v_mfma_f32_16x16x4_f32, W and source
    tiles by LDS-DMA, batch-major state) and the r02 form "mfma32" (v_mfma_f32_32x32x2_f32, neuron-major [n, 32] state).
    ``v`` / ``source`` are [B, rows] / [B, n] views in every form.  Rate models, n % 4 == 0.
    """

    BATCHES = (2, 4, 8, 16)   # vector-ALU form
    SLOTS = _abi.RP_BATCH_SLOTS
    FORMS = {"valu": _abi.RP_BATCH_VALU, "mfma32": _abi.RP_BATCH_MFMA, "mfma16": _abi.RP_BATCH_MFMA16}
    DEFAULT_MFMA = "mfma16"   # which matrix-core form `mfma=True` / batch > 8 selects

    def __init__(self, pop: Population, batch: int, v0=None, mfma: Optional[bool] = None, form: Optional[str] = None,
                 exact: bool = False):
        """``form``: "valu" | "mfma16" (v_mfma_f32_16x16x4_f32, LDS-DMA ring, batch-major state) | "mfma32" (the r02
        v_mfma_f32_32x32x2_f32 form, neuron-major state); default: "valu" up to 4 trajectories (bit-identical to unbatched
        runs at the HBM roof), ``DEFAULT_MFMA`` from 5 (N = 32768, 8 trajectories: 0.63 ms per step against 0.87 on the
        vector ALUs).  ``exact=True`` keeps 8 (and 16) trajectories on the vector ALUs, i.e. bit-identical to unbatched
        runs; ``mfma=True`` asks for the matrix cores at any batch."""
        if pop.spiking:
            raise NotImplementedError("batched trajectories cover the rate models (li_tanh, li_sigmoid)")
        if pop.n % 4 or pop.W.stride(0) % 4:
            raise ValueError("batched trajectories need n % 4 == 0")
        if form is None:
            if exact and mfma is None and batch > 4 and batch not in self.BATCHES:
                raise ValueError(f"exact=True (bit-identical to unbatched runs) needs a vector-ALU batch size {self.BATCHES} or at most 4 "
                                 f"trajectories, got {batch} (Network.run_batch pads to the next one)")
            use_mfma = (batch > 4 and not (exact and batch in self.BATCHES)) if mfma is None else bool(mfma)
            form = self.DEFAULT_MFMA if use_mfma else "valu"
            # more than 16 trajectories on a SMALL population: the r02 form's 32-row workgroups fill the chip where the
            # 64-row workgroups of "mfma16" cannot (scripts/archive/sweep_mfma16.py: N = 8192, B = 32: 0.075 vs 0.096 ms per step;
            # N = 16384: 0.227 vs 0.206)
            if use_mfma and batch > 16 and pop.rows < 12288 and pop.row0 % 32 == 0:
                form = "mfma32"
            # a shard offset that is a multiple of 32 but not of 128: only the r02 form can take it
            if use_mfma and form == "mfma16" and pop.row0 % 128 and pop.row0 % 32 == 0:
                form = "mfma32"
        if form not in self.FORMS:
            raise ValueError(f"form must be one of {sorted(self.FORMS)}, got {form!r}")
        self.form = form
        self.mfma = form != "valu"
        self.neuron_major = form == "mfma32"
        if self.mfma and not 1 <= batch <= self.SLOTS:
            raise ValueError(f"the matrix-core forms hold 1..{self.SLOTS} trajectories, got {batch}")
        if not self.mfma and batch not in self.BATCHES:
            raise ValueError(f"batch must be one of {self.BATCHES} (vector-ALU form) or <= {self.SLOTS} (mfma=True), got {batch}")
        align = {"mfma32": 32, "mfma16": 128}.get(form, 8)
        if pop.row0 % align:
            raise ValueError(f"the '{form}' form needs a shard offset that is a multiple of {align}")
        self.pop, self.batch = pop, int(batch)
        f32 = dict(dtype=torch.float32, device=pop.device)
        if self.neuron_major:
            self._v = torch.zeros(pop.rows, self.SLOTS, **f32)
            self._src = [torch.zeros(pop.n, self.SLOTS, **f32), torch.zeros(pop.n, self.SLOTS, **f32)]
        else:
            self._v = torch.zeros(batch, pop.rows, **f32)
            self._src = [torch.zeros(batch, pop.n, **f32), torch.zeros(batch, pop.n, **f32)]
        self.cur = 0
        if v0 is not None:
            self.v.copy_(torch.as_tensor(v0, dtype=torch.float32).to(pop.device).expand(batch, pop.rows))
        self.refresh()

    @property
    def v(self) -> torch.Tensor:
        """States [B, rows] (a view: writes go to the device state; call :meth:`refresh` afterwards)."""
        return self._v.t()[:self.batch] if self.neuron_major else self._v

    @property
    def source(self) -> torch.Tensor:
        """Current source vectors r_b = act(v_b), [B, n] (view)."""
        s = self._src[self.cur]
        return s.t()[:self.batch] if self.neuron_major else s

    def refresh(self):
        """r_b = act(v_b) after the states were set from the host (through the unbatched refresh kernel, row by row)."""
        pop = self.pop
        keep_v, keep_src, keep_cur = pop.v, pop.src, pop.cur
        try:
            for b in range(self.batch):
                vb = self.v[b].contiguous()
                sb = torch.zeros(pop.n, dtype=torch.float32, device=pop.device)
                pop.v, pop.src, pop.cur = vb, [sb, sb.clone()], 0
                pop._net_cache = None
                pop.refresh()
                self.source[b, pop.row0:pop.row0 + pop.rows] = sb[pop.row0:pop.row0 + pop.rows]
        finally:
            pop.v, pop.src, pop.cur = keep_v, keep_src, keep_cur
            pop._net_cache = None

    def reset_sampling(self):
        self.acc_count = 0  # an empty buffer: the library overwrites `acc` with the first buffered output

    def forward(self, steps: int, x: Optional[torch.Tensor] = None, traj: Optional[torch.Tensor] = None,
                out_var: int = _abi.RP_OUT_STATE0, w_in: Optional[torch.Tensor] = None, step0: int = 0, cutoff: int = 0,
                sampling_steps: int = 1, samples: Optional[torch.Tensor] = None, rec: Optional[dict] = None,
                sample_idx: int = 0) -> int:
        """`steps` Euler steps of all trajectories; returns the next sample index.
        x: None | [steps, B, rows] | [steps, B, 1] | [steps, B, m] together with ``w_in`` [rows, m] (the input edge inside
        the step); traj: optional [steps, B, rows] receiving the pre-update outputs; samples / rec[0 | 1]: optional
        [B, S, rows] buffers receiving, per trajectory, what :meth:`Population.forward` samples for one (buffer means of the
        output, post-update v / r at the sample steps; ``step0`` / ``cutoff`` / ``sampling_steps`` / ``sample_idx`` as there)."""
        pop, B = self.pop, self.batch
        bt = _abi.RpBatch()
        bt.batch, bt.steps = B, int(steps)
        bt.layout = self.FORMS[self.form]
        bt.v, bt.ldv = self._v.data_ptr(), self._v.stride(0)
        bt.src[0], bt.src[1], bt.ldsrc = self._src[0].data_ptr(), self._src[1].data_ptr(), self._src[0].stride(0)
        bt.cur = self.cur
        keep = []
        if x is None:
            bt.x_mode = _abi.RP_X_NONE
        else:
            if x.dim() != 3 or x.shape[0] < steps or x.shape[1] != B:
                raise RuntimeError(f"inputs must be [>= {steps}, {B}, width], got {tuple(x.shape)}")
            x = x.detach().to(dtype=torch.float32, device=pop.device).contiguous()
            keep.append(x)
            bt.x, bt.ldx = x.data_ptr(), x.shape[2]
            if w_in is not None:
                w_in = w_in.detach().to(dtype=torch.float32, device=pop.device).contiguous()
                if tuple(w_in.shape) != (pop.rows, x.shape[2]):
                    raise RuntimeError(f"input weights {tuple(w_in.shape)} do not map {x.shape[2]} inputs to {pop.rows} neurons")
                keep.append(w_in)
                bt.x_mode, bt.m, bt.w_in, bt.ldwin = _abi.RP_X_PROJ, x.shape[2], w_in.data_ptr(), w_in.stride(0)
            elif x.shape[2] == pop.rows and pop.rows != 1:
                bt.x_mode = _abi.RP_X_FULL
            elif x.shape[2] == 1:
                bt.x_mode = _abi.RP_X_BCAST
            else:
                raise RuntimeError(f"inputs must be [>= {steps}, {B}, {pop.rows} or 1], got {tuple(x.shape)}")
        bt.out_var = int(out_var)
        if traj is not None:
            if traj.dtype != torch.float32 or traj.device != pop.device or traj.dim() != 3 or not traj.is_contiguous() \
                    or traj.shape[0] < steps or traj.shape[1] != B or traj.shape[2] < pop.rows:
                raise RuntimeError(f"traj must be a contiguous float32 [>= {steps}, {B}, >= {pop.rows}] tensor on {pop.device}")
            bt.traj, bt.ldtraj = traj.data_ptr(), traj.shape[2]
        bufs = [("samples", samples)] + [(f"rec[{k}]", v) for k, v in (rec or {}).items()]
        bufs = [(name, t) for name, t in bufs if t is not None]
        if bufs:
            shape = None
            for name, t in bufs:
                if t.dtype != torch.float32 or t.device != pop.device or t.dim() != 3 or not t.is_contiguous() \
                        or t.shape[0] != B or t.shape[2] < pop.rows or (shape is not None and tuple(t.shape) != shape):
                    raise RuntimeError(f"{name} must be a contiguous float32 [{B}, S, >= {pop.rows}] tensor on {pop.device} "
                                       "(all sample / record buffers of one shape)")
                shape = tuple(t.shape)
            if getattr(self, "_acc", None) is None:
                self._acc = torch.empty(B, pop.rows, dtype=torch.float32, device=pop.device)
                self.acc_count = 0
            bt.step0, bt.cutoff, bt.sampling_steps, bt.max_samples = int(step0), int(cutoff), int(sampling_steps), shape[1]
            bt.acc, bt.ldacc, bt.acc_count, bt.sample_idx = self._acc.data_ptr(), self._acc.stride(0), self.acc_count, int(sample_idx)
            bt.samples, bt.ldsamples = _ptr(samples), shape[2]
            for k, t in (rec or {}).items():
                if int(k) not in (0, 1):
                    raise RuntimeError("the batched step records v (0) and r (1)")
                bt.rec[int(k)] = _ptr(t)
        pop.backend.forward_batched(pop._net(), bt, pop.device)
        self.cur = bt.cur
        if bufs:
            self.acc_count = bt.acc_count
        del keep
        return int(bt.sample_idx)
