"""Row-sharded integration of one population over the GPUs of a node.

Rank p owns rows [p*n/P, (p+1)*n/P) of W (target neurons) and the matching slices of every
per-neuron vector.  Each Euler step is: local fused step kernel (writes the rank's slice of
the next source vector r = act(v) / s) -> all-gather of the slices into the full source vector
(n floats in total) -> next step.  The collective is `torch.distributed` (backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).  The reference has no distributed code;
the oracle for this path is the single-device result, which the sharded run reproduces bit for
bit because the per-row summation tree does not depend on the sharding.
"""
from typing import Dict, Optional

import torch
import torch.distributed as dist

from . import _abi
from .engine import Population


def rccl_comm(group, device, backend) -> Optional[_abi.RpComm]:
    """The group's RCCL communicator as an ``rp_comm`` for the in-library exchange (rp_forward_sharded /
    rp_backward_sharded), or None when the exchange has to go through ``torch.distributed`` (gloo group, the
    CPU test backend, or RECTIPY_AMD_HOST_COLLECTIVES=1).  The communicator stays owned by ProcessGroupNCCL; the
    function addresses come from the RCCL instance PyTorch loaded, the only one in which that communicator is valid."""
    import ctypes
    import os
    import warnings
    if os.environ.get("RECTIPY_AMD_HOST_COLLECTIVES", "0") == "1" or not hasattr(backend, "forward_sharded"):
        return None
    if dist.get_backend(group) != "nccl":
        return None
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    try:
        ptr = pg._get_backend(torch.device(device))._comm_ptr()  # private torch API: may be absent in other builds
        if not ptr:
            raise RuntimeError("communicator not initialised on " + str(device))
        lib = _loaded_rccl()
        comm = _abi.RpComm()
        comm.comm = ptr
        comm.all_gather = ctypes.cast(lib.ncclAllGather, ctypes.c_void_p).value
        comm.reduce_scatter = ctypes.cast(lib.ncclReduceScatter, ctypes.c_void_p).value
    except (AttributeError, OSError, RuntimeError) as e:
        global _warned_fallback
        if not _warned_fallback:
            warnings.warn(f"in-library RCCL exchange unavailable ({type(e).__name__}: {e}); the per-step exchange goes "
                          "through torch.distributed from the host loop (same results, higher per-step overhead)")
            _warned_fallback = True
        return None
    comm.rank, comm.world = dist.get_rank(group), dist.get_world_size(group)
    return comm


_warned_fallback = False


def agreed_comm(comm, group, device):
    """`comm` if EVERY rank of the group obtained one, else None on every rank (collective; `rccl_comm` can fail on a
    single rank, and a rank in the host loop next to ranks that enqueue the in-library collective would hang the job)."""
    have = torch.tensor([int(comm is not None)], device=device)
    dist.all_reduce(have, op=dist.ReduceOp.MIN, group=group)
    return comm if int(have.item()) == 1 else None


def _loaded_rccl():
    """The RCCL instance that owns PyTorch's communicators: torch's bundled lib/librccl.so when it ships one (dlopen
    returns the already-loaded instance), else the symbols visible in the process (torch linked against a system
    RCCL)."""
    import ctypes
    import os
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if os.path.exists(bundled):
        return ctypes.CDLL(bundled)
    lib = ctypes.CDLL(None)
    lib.ncclAllGather, lib.ncclReduceScatter  # AttributeError when not visible -> host collectives
    return lib


class DirectExchange:
    """``rp_comm`` for the stream-ordered DIRECT exchange (``RP_COMM_DIRECT``, include/rectipy_hip.h): the population's
    two source buffers and a flag array live in device memory that the other rank processes map through hipIpc; every
    step a rank pushes its slice into its peers' buffers and the STREAMS signal / wait on per-step sequence numbers
    (hipStreamWriteValue32 / hipStreamWaitValue32) -- no collective launch and no spinning kernel.  Opt-in
    (``ShardedPopulation(..., exchange="direct")`` or RECTIPY_AMD_EXCHANGE=direct): it has been run with two processes
    on ONE MI355X (tests/test_direct_exchange.py: bit-equal to the host-loop exchange), never over xGMI.  Forward: the
    all-gather of the source slices; adjoint (r04): every rank pushes the owners' slices of its full-length partial
    W_p^T u_p into their inboxes and the owner adds them in RANK ORDER -- a cross-rank sum that is the same from run to run
    (an all-reduce / reduce-scatter promises no order).  Collective constructor: every rank of `group` calls it."""

    def __init__(self, pop: Population, group=None):
        import ctypes as C
        from . import _lib
        self.lib = _lib.load()
        self.group = group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if pop.rows % 4 or pop.rows * world != pop.n:
            raise ValueError("the direct exchange needs n = world * rows and rows % 4 == 0")
        n, dev = pop.n, pop.device
        self._own, self._mapped = [], []
        handles = []
        with torch.cuda.device(dev):
            # two source buffers, the flag array, the inbox of the sharded adjoint (2 parities x n floats)
            for nbytes in (4 * n, 4 * n, 4 * max(world, 1), 8 * n):
                ptr, h = C.c_void_p(), C.create_string_buffer(64)
                _lib.check(self.lib.rp_ipc_alloc(nbytes, C.byref(ptr), h))
                self._own.append(ptr.value)
                handles.append(h.raw)
        # the population's source buffers move into the exportable allocations (same values)
        new_src = [_wrap_device_floats(self._own[k], n, dev, self) for k in range(2)]
        for k in range(2):
            new_src[k].copy_(pop.src[k])
        pop.src = new_src
        pop._net_cache = None
        everyone = [None] * world
        dist.all_gather_object(everyone, handles, group=group)
        self.peer_src = (C.c_void_p * (2 * world))()
        self.peer_flags = (C.c_void_p * world)()
        self.peer_inbox = (C.c_void_p * world)()
        with torch.cuda.device(dev):
            for p in range(world):
                if p == rank:
                    ptrs = self._own
                else:
                    ptrs = []
                    for h in everyone[p]:
                        q = C.c_void_p()
                        _lib.check(self.lib.rp_ipc_open(h, C.byref(q)))
                        ptrs.append(q.value)
                        self._mapped.append(q.value)
                self.peer_src[2 * p], self.peer_src[2 * p + 1], self.peer_flags[p] = ptrs[0], ptrs[1], ptrs[2]
                self.peer_inbox[p] = ptrs[3]
        comm = _abi.RpComm()
        comm.rank, comm.world, comm.mode, comm.seq = rank, world, _abi.RP_COMM_DIRECT, 0
        comm.peer_src = C.cast(self.peer_src, C.POINTER(C.c_void_p))
        comm.peer_flags = C.cast(self.peer_flags, C.POINTER(C.c_void_p))
        comm.flags = self._own[2]
        comm.peer_inbox = C.cast(self.peer_inbox, C.POINTER(C.c_void_p))
        comm.inbox = self._own[3]
        self.comm = comm
        # the own allocations are released when this object -- and with it every tensor that wraps them -- is gone
        import weakref
        self._finalizer = weakref.finalize(self, _release_ipc, self.lib, self._own, self._mapped)
        dist.barrier(group=group)  # nobody pushes before everybody has mapped

    def close(self):
        """Unmap the peers' buffers (collective by convention: call it on every rank once nobody pushes any more).  The own
        allocations stay alive with the tensors that wrap them and are freed when those are gone.

        The protocol itself has no time-out: a rank that dies in the middle of a run leaves its peers' streams waiting for
        a sequence number that never comes (hipStreamWaitValue32 has no deadline).  ShardedPopulation therefore runs a
        host-side StallWatchdog next to it: no progress of the peers' sequence numbers for `deadline_s` seconds while work
        is outstanding ends the rank with a non-zero exit code."""
        from . import _lib
        torch.cuda.synchronize()
        if getattr(self, "_watchdog", None) is not None:
            self._watchdog.stop()
        mapped, self._mapped[:] = list(self._mapped), []
        for q in mapped:
            _lib.check(self.lib.rp_ipc_close(q), self.lib)


def _release_ipc(lib, own, mapped):
    """Finalizer of a DirectExchange: unmap what is still mapped, free the own allocations (errors are not raised from a
    finalizer: the process is usually shutting down)."""
    try:
        for q in list(mapped):
            lib.rp_ipc_close(q)
        for ptr in own:
            lib.rp_ipc_free(ptr)
    except Exception:
        pass


def _wrap_device_floats(ptr: int, count: int, device, owner) -> torch.Tensor:
    """A float32 torch tensor over `count` floats of device memory at `ptr` (not owned by torch; `owner` is kept alive by
    the tensor through the array-interface object)."""
    class _Mem:
        pass

    mem = _Mem()
    mem.owner = owner
    mem.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(mem, device=device)


class StallWatchdog:
    """Host-side deadline for the per-step exchange of a row-sharded run.  The stream-ordered direct exchange waits for
    its peers with hipStreamWaitValue32, which has no time-out, and a host-loop collective blocks in the process group: a
    rank that dies or hangs would leave its peers waiting for ever.  A daemon thread watches a PROGRESS number (the sum of
    the rank's flag array -- the peers' sequence numbers -- read on a side stream; the host loop's step counter) while the
    rank is BUSY (work outstanding on the compute stream / inside the host loop): no progress for `deadline_s` seconds ends
    the process with exit code ``EXIT_CODE`` and a message on stderr, so that the launcher sees a failed rank instead of a
    silent hang.  ``RECTIPY_AMD_EXCHANGE_DEADLINE`` (seconds; 0 = off) sets the default of ShardedPopulation."""

    EXIT_CODE = 75   # EX_TEMPFAIL

    def __init__(self, deadline_s: float, progress, busy, what: str = "exchange"):
        import threading
        self.deadline_s, self._progress, self._busy, self.what = float(deadline_s), progress, busy, what
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._watch, name="rectipy_amd-stall-watchdog", daemon=True)
        self._thread.start()
        import atexit
        atexit.register(self.stop)   # (no device queries while the interpreter shuts down)

    def stop(self):
        self._stop.set()

    def _watch(self):
        import os
        import sys
        import time
        last, since = None, time.monotonic()
        poll = max(0.05, min(1.0, self.deadline_s / 4))
        while not self._stop.wait(poll):
            try:
                if not self._busy():
                    last, since = None, time.monotonic()
                    continue
                now = self._progress()
            except Exception:
                continue   # (interpreter shutting down, device gone: nothing to judge)
            if now != last:
                last, since = now, time.monotonic()
            elif time.monotonic() - since > self.deadline_s:
                sys.stderr.write(f"[rectipy_amd] rank stalled: no progress of the {self.what} for {self.deadline_s:.0f} s "
                                 f"(a peer rank died or hangs); exiting with code {self.EXIT_CODE}\n")
                sys.stderr.flush()
                os._exit(self.EXIT_CODE)


def one_rank_env(environ=None):
    """Rendezvous variables for a process started WITHOUT a launcher (plain `python`): a one-rank group on 127.0.0.1
    at a free port.  Only the variables that are missing are filled in -- a partially set environment (a launcher's
    MASTER_ADDR / WORLD_SIZE) is never overwritten -- and no port is hard-coded, so two such runs on one host do not
    collide."""
    import os
    import socket
    env = os.environ if environ is None else environ
    if "MASTER_PORT" not in env:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(s.getsockname()[1])
    for key, val in (("MASTER_ADDR", "127.0.0.1"), ("RANK", "0"), ("WORLD_SIZE", "1")):
        env.setdefault(key, val)
    return env


def shard_bounds(n: int, world: int, rank: int):
    if n % world:
        raise ValueError(f"n={n} must be divisible by the number of ranks ({world})")
    rows = n // world
    return rank * rows, rows


class ShardedPopulation:
    """One rank's shard plus the per-step exchange."""

    def __init__(self, model: str, weights_local, params_local: Dict[str, object], dt: float, device, n: int,
                 v0_local=None, s0_local=None, group=None, backend=None, exchange: Optional[str] = None,
                 deadline_s: Optional[float] = None, **kw):
        """``exchange``: None / "rccl" -- the per-step RCCL all-gather enqueued in-library (host loop for gloo groups);
        "direct" -- the stream-ordered push exchange (:class:`DirectExchange`; opt-in, also RECTIPY_AMD_EXCHANGE=direct).
        ``deadline_s``: seconds without progress of the exchange after which the rank exits (:class:`StallWatchdog`;
        default RECTIPY_AMD_EXCHANGE_DEADLINE or 120, 0 = off; direct exchange and host loop -- a rank lost inside an RCCL
        collective is the launcher's and RCCL's own watchdog's to end)."""
        import os
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.row0, self.rows = shard_bounds(n, self.world, self.rank)
        self.pop = Population(model, weights_local, params_local, dt, device, v0=v0_local, s0=s0_local, n=n,
                              row0=self.row0, backend=backend, **kw)
        self.n = n
        self._gather_source()  # every rank needs the full initial source vector
        exchange = exchange or os.environ.get("RECTIPY_AMD_EXCHANGE", "rccl")
        self.direct = None
        if exchange == "direct":
            if not hasattr(self.pop.backend, "forward_sharded") or self.pop.device.type != "cuda":
                raise RuntimeError("the direct exchange runs on the HIP backend only")
            self.direct = DirectExchange(self.pop, group)
            self.comm = self.direct.comm
        elif exchange == "rccl":
            self.comm = agreed_comm(rccl_comm(group, self.pop.device, self.pop.backend), group, self.pop.device)
        else:
            raise ValueError(f"exchange must be 'rccl' or 'direct', got {exchange!r}")
        # a host-side deadline for a dead or hanging peer (StallWatchdog): the direct exchange and the host loop have none
        self._beats, self._in_loop = 0, False
        deadline = float(os.environ.get("RECTIPY_AMD_EXCHANGE_DEADLINE", "120")) if deadline_s is None else float(deadline_s)
        self.watchdog = None
        if deadline > 0 and self.world > 1:
            if self.direct is not None:
                dev = self.pop.device
                flags = _wrap_device_floats(self.direct._own[2], self.world, dev, self.direct).view(torch.int32)
                side = torch.cuda.Stream(device=dev)
                main = torch.cuda.current_stream(dev)

                def progress():
                    with torch.cuda.stream(side):
                        return int(flags.sum().item())
                self.watchdog = StallWatchdog(deadline, progress, lambda: not main.query(), "direct exchange (peer sequence numbers)")
                self.direct._watchdog = self.watchdog   # (stopped by DirectExchange.close: nothing is pushed afterwards)
            elif self.comm is None:
                self.watchdog = StallWatchdog(deadline, lambda: self._beats, lambda: self._in_loop, "host-loop exchange")

    def _gather_source(self):
        full = self.pop.source
        mine = full[self.row0:self.row0 + self.rows].clone()
        if full.is_cuda and dist.get_backend(self.group) != "nccl":
            # a CPU-only process group around GPU ranks (the two-processes-on-one-GPU test of the direct exchange)
            parts = [None] * self.world
            dist.all_gather_object(parts, mine.cpu(), group=self.group)
            full.copy_(torch.cat(parts).to(full.device))
            return
        dist.all_gather_into_tensor(full, mine, group=self.group)

    def check_inlib_exchange(self, steps: int = 3) -> bool:
        """One-time guard for the in-library RCCL exchange (it runs on the compute stream with ProcessGroupNCCL's
        communicator, outside the process group's own bookkeeping): `steps` steps with it must reproduce, bit for bit,
        the same steps with the ``torch.distributed`` all-gather from the host loop.  On disagreement or a library
        error on ANY rank, every rank drops to the host loop (``self.comm = None``).  State is restored.  Collective:
        all ranks must call it."""
        return check_inlib_exchange(self.pop, self, steps, self.group)

    def forward(self, steps: int, x: Optional[torch.Tensor] = None, out_var: int = _abi.RP_OUT_STATE0,
                traj: Optional[torch.Tensor] = None, step0: int = 0, **sampling):
        """`steps` Euler steps; `x` and `traj` are LOCAL ([steps, rows] or [steps, 1])."""
        idx = sampling.pop("sample_idx", 0)
        if self.comm is not None:  # one library call: step kernels and RCCL all-gathers back to back on one stream
            return self.pop.forward(steps, x=x, out_var=out_var, traj=traj, step0=step0, sample_idx=idx,
                                    comm=self.comm, **sampling)
        self._in_loop = True
        try:
            for t in range(steps):
                xt = None if x is None else x[t:t + 1]
                tr = None if traj is None else traj[t:t + 1]
                idx = self.pop.forward(1, x=xt, out_var=out_var, traj=tr, step0=step0 + t, sample_idx=idx, **sampling)
                full = self.pop.source  # the buffer the kernel just wrote this rank's slice into
                dist.all_gather_into_tensor(full, full[self.row0:self.row0 + self.rows], group=self.group)
                self._beats += 1
        finally:
            self._in_loop = False
        return idx

    def backward(self, traj_v, traj_r, grad_out, grad_var, lam, u_traj, **kw):
        """Adjoint sweep over a stored window of this rank's shard (arguments of :meth:`Population.backward`, all local):
        per step the full-length partial W_p^T u_p, its cross-rank sum -- RCCL reduce-scatter enqueued in-library, the
        direct exchange's rank-ordered sum (``exchange="direct"``), or the host loop -- and the local update."""
        return self.pop.backward(traj_v, traj_r, grad_out, grad_var, lam, u_traj, exchange=_Exchange(self), **kw)


def simulate_sharded_backward(shards: list, windows: list) -> None:
    """The sharded adjoint sweep of `len(shards)` row shards of ONE population evaluated in ONE process: per step every
    shard's full-length partial W_p^T u_p (rp_bwd_partial), the partials' slices added in RANK ORDER 0, 1, ... (the fixed
    cross-rank order of the direct exchange and of the host loop, rp_comm.peer_inbox), the owners' local updates
    (rp_bwd_apply).  `shards`: the ranks' :class:`Population` objects in rank order; `windows`: per rank the keyword
    arguments of :meth:`Population.backward` (local trajectories, adjoint state and gradient buffers, updated in place).
    This is the mirror the multi-process runs are compared with (tests: on the oracle backend)."""
    world = len(shards)
    n, rows = shards[0].n, shards[0].rows
    dev = shards[0].device
    T = int(windows[0]["traj_v"].shape[0])

    z_full = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(world)]
    # Population.backward drives the whole sweep of ONE rank; here the ranks advance in lock step, so the sweep is
    # restated with the step-wise entry points
    prepared = [pop._backward_args(**kw) for pop, kw in zip(shards, windows)]
    for pop, (net, b, keep) in zip(shards, prepared):
        pop.backend.bwd_prepare(net, b, pop.device)
    for t in range(T - 1, -1, -1):
        for p, (pop, (net, b, keep)) in enumerate(zip(shards, prepared)):
            pop.backend.bwd_partial(net, b, t, z_full[p].data_ptr(), pop.device)
        for q, (pop, (net, b, keep)) in enumerate(zip(shards, prepared)):
            total = z_full[0][q * rows:(q + 1) * rows].clone()
            for p in range(1, world):
                total = total + z_full[p][q * rows:(q + 1) * rows]
            pop.backend.bwd_apply(net, b, t, total.contiguous().data_ptr(), pop.device)


def check_inlib_exchange(pop: Population, holder, steps: int, group) -> bool:
    """See ShardedPopulation.check_inlib_exchange; `holder.comm` is the rp_comm in use (set to None on failure).

    Collective.  The ranks first agree -- through the process group's own all-reduce -- on whether EVERY rank holds a
    communicator (`rccl_comm` can fail on one rank only), so that no rank ever enqueues the in-library all-gather while
    another one has already left for the host loop.  A library error during the comparison is not survivable: some
    ranks may have enqueued a collective the failing rank never will, so the error propagates (the launcher ends the
    job) instead of continuing with collectives on that communicator."""
    holder.comm = agreed_comm(holder.comm, group, pop.device)
    if holder.comm is None:
        return False
    saved = (pop.v.clone(), pop.src[0].clone(), pop.src[1].clone(), pop.cur,
             None if pop.aux is None else pop.aux.clone(), None if pop.aux2 is None else pop.aux2.clone())
    row0, rows = pop.row0, pop.rows

    def restore():
        pop.v.copy_(saved[0]); pop.src[0].copy_(saved[1]); pop.src[1].copy_(saved[2]); pop.cur = saved[3]
        if saved[4] is not None:
            pop.aux.copy_(saved[4])
        if saved[5] is not None:
            pop.aux2.copy_(saved[5])

    def run(comm):
        restore()
        if comm is not None:
            pop.forward(steps, comm=comm)
        else:
            for _ in range(steps):
                pop.forward(1)
                full = pop.source
                dist.all_gather_into_tensor(full, full[row0:row0 + rows], group=group)
        return pop.v.clone(), pop.source.clone()

    a, b = run(holder.comm), run(None)
    agree = int(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))
    same = torch.tensor([agree], device=pop.device)
    dist.all_reduce(same, op=dist.ReduceOp.MIN, group=group)
    restore()
    if int(same.item()) != 1:
        holder.comm = None
        return False
    return True


# ---- front end ---------------------------------------------------------------------------------------------

import types

import numpy as np

from .network import Network
from .nodes import RateNet


def _slice_rows(val, row0, rows, n):
    """Per-neuron vector / matrix rows of the global network -> this rank's slice (scalars pass through)."""
    if isinstance(val, (int, float)):
        return val
    arr = val if isinstance(val, torch.Tensor) else np.asarray(val)
    if arr.ndim >= 1 and arr.shape[0] == n:
        return arr[row0:row0 + rows]
    return val


class _AllReduceSum(torch.autograd.Function):
    """Sum of per-rank partial read-outs.  Every rank evaluates the same loss on the summed prediction, so the gradient
    with respect to a rank's partial sum is the gradient of the total: backward is the identity (no second collective)."""

    @staticmethod
    def forward(ctx, part, group):
        total = part.detach().clone().contiguous()
        dist.all_reduce(total, group=group)
        return total

    @staticmethod
    def backward(ctx, g):
        return g, None


class _ShardedRateWindow(torch.autograd.Function):
    """T Euler steps of a row-sharded leaky-integrator population as one autograd node.

    forward : per step  local fused step kernel -> all-gather of the source vector (and a copy of the full r_t for dW);
    backward: per step  column-strip kernel over the rank's rows (full-length partial W_p^T u_p) -> reduce-scatter ->
              local adjoint update (rp_bwd_prepare / rp_bwd_partial / rp_bwd_apply);
              dW_p = dt U_p^T R (rows x n) is local: one rectangular MFMA GEMM per window."""

    @staticmethod
    def forward(ctx, node, net, steps, x, v_in, W, tau, k, eta, *act):
        pop = node._pop
        rows, n, dev = pop.rows, pop.n, pop.device
        row0 = pop.row0
        f32 = dict(dtype=torch.float32, device=dev)
        traj_v = torch.empty(steps, rows, **f32)
        traj_r = torch.empty(steps, rows, **f32)
        traj_r_full = torch.empty(steps, n, **f32)
        traj_acc = torch.empty(steps, rows, **f32) if k.requires_grad else None
        if net.comm is not None:  # whole window in one library call (step kernels + RCCL all-gathers on one stream)
            pop.forward(steps, x=x, save_v=traj_v, save_src=traj_r, save_acc=traj_acc, save_src_full=traj_r_full,
                        comm=net.comm)
        else:
            for t in range(steps):
                traj_r_full[t].copy_(pop.source)
                pop.forward(1, x=None if x is None else x[t:t + 1], save_v=traj_v[t:t + 1], save_src=traj_r[t:t + 1],
                            save_acc=None if traj_acc is None else traj_acc[t:t + 1])
                net._gather(node, row0, rows)
        ctx.node, ctx.net, ctx.steps = node, net, steps
        ctx.x_width = None if x is None else x.shape[1]
        ctx.saved = (traj_v, traj_r, traj_r_full, traj_acc)
        node._last_window = (traj_v, traj_r)
        out = traj_r if node._out_code == _abi.RP_OUT_SOURCE else traj_v
        return out.clone(), pop.v.clone()

    @staticmethod
    def backward(ctx, g_traj, g_vend):
        node, net, T = ctx.node, ctx.net, ctx.steps
        pop = node._pop
        rows, n, dev = pop.rows, pop.n, pop.device
        traj_v, traj_r, traj_r_full, traj_acc = ctx.saved
        f32 = dict(dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad  # (node, net, steps, x, v_in, W, tau, k, eta, *act)
        act_names = ("r_max", "s", "v0")[:len(need) - 9]
        g_act = {q: torch.zeros(rows, **f32) for i, q in enumerate(act_names) if need[9 + i]}
        lam = torch.zeros(rows, **f32) if g_vend is None else g_vend.detach().to(**f32).clone().contiguous()
        g_out = None if g_traj is None else g_traj.detach().to(**f32).contiguous()
        u_traj = torch.empty(T, rows, **f32)
        g_tau = torch.zeros(rows, **f32) if need[6] else None
        g_k = torch.zeros(rows, **f32) if need[7] else None
        g_eta = torch.zeros(rows, **f32) if need[8] else None
        g_x = torch.empty(T, rows, **f32) if need[3] else None
        keep = pop.backward(traj_v, traj_r, g_out, node._out_code, lam, u_traj, traj_acc=traj_acc, grad_tau=g_tau,
                            grad_k=g_k, grad_eta=g_eta, grad_x=g_x, grad_params=g_act, exchange=_Exchange(net))
        g_W = None
        if need[5]:
            g_W = torch.empty(rows, n, **f32)
            pop.backend.outer_store_rect(rows, n, T, float(pop.dt), u_traj.data_ptr(), rows, traj_r_full.data_ptr(), n,
                                         g_W.data_ptr(), n, dev)
        if g_x is not None and ctx.x_width == 1:
            g_x = g_x.sum(dim=1, keepdim=True)
            dist.all_reduce(g_x, group=net.group)  # a broadcast input feeds every shard
        del keep
        return (None, None, None, g_x, (lam if need[4] else None), g_W, g_tau, g_k, g_eta) + \
            tuple(g_act.get(q) for q in act_names)


class _Exchange:
    """Cross-rank sum of the full-length partials of the sharded adjoint: in-library (``comm``) or one
    ``torch.distributed`` reduce-scatter per step."""

    def __init__(self, net):
        self.comm = net.comm
        self.group = net.group
        self.nccl = dist.get_backend(net.group) == "nccl"

    def reduce_scatter(self, z_full, z_local):
        if self.nccl:
            dist.reduce_scatter_tensor(z_local, z_full, group=self.group)
            return
        # host loop (gloo; CPU tests, GPU ranks around a CPU group): every rank's contribution to this rank's slice,
        # added in RANK ORDER -- the fixed order of the direct exchange (include/rectipy_hip.h: rp_comm.peer_inbox), so
        # that the host loop, the direct exchange and the oracle's mirror agree bit for bit
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        rows = z_local.numel()
        send = [z_full[q * rows:(q + 1) * rows].detach().cpu().contiguous() for q in range(world)]
        recv = [torch.empty(rows) for _ in range(world)]
        dist.all_to_all(recv, send, group=self.group) if dist.get_backend(self.group) != "gloo" else self._gloo_all_to_all(recv, send)
        total = recv[0]
        for p in range(1, world):
            total = total + recv[p]
        z_local.copy_(total.to(z_local.device))

    def _gloo_all_to_all(self, recv, send):
        """gloo has no all_to_all for CPU tensors in every build: P gathers of one slice each."""
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        for q in range(world):
            dist.gather(send[q], recv if q == rank else None, dst=dist.get_global_rank(self.group, q) if self.group else q,
                        group=self.group)


class _ShardedSpikeWindow(torch.autograd.Function):
    """T steps of a row-sharded spiking population (QIF, QIF_SFA, LIF, IK) as one autograd node: forward as
    `_ShardedRateWindow` with the all-gather of s; backward = surrogate-gradient adjoint on the rank's neurons with the
    reduce-scatter of W_p^T u_p per step, dW_p = dt U_p^T S one rectangular MFMA GEMM."""

    @staticmethod
    def forward(ctx, node, net, steps, x, y_in, W, *params):
        from .spiking_autograd import TRAINABLE
        pop = node._pop
        rows, n, dev = pop.rows, pop.n, pop.device
        f32 = dict(dtype=torch.float32, device=dev)
        names = TRAINABLE[pop.model]
        req = {q: p.requires_grad for q, p in zip(names, params)}
        need_acc = pop.model == "ik" or req.get("k", False)
        traj_v, traj_s = torch.empty(steps, rows, **f32), torch.empty(steps, rows, **f32)
        traj_aux = torch.empty(steps, rows, **f32) if pop.aux is not None else None
        traj_acc = torch.empty(steps, rows, **f32) if need_acc else None
        traj_s_full = torch.empty(steps, n, **f32)
        kw = dict(x_target=node._x_target, spike_events=node.spike_events, spike_count=node._spike_count)
        if net.comm is not None:
            pop.forward(steps, x=x, save_v=traj_v, save_src=traj_s, save_aux=traj_aux, save_acc=traj_acc,
                        save_src_full=traj_s_full, comm=net.comm, **kw)
        else:
            row = lambda t_, i: None if t_ is None else t_[i:i + 1]
            for t in range(steps):
                traj_s_full[t].copy_(pop.source)
                pop.forward(1, x=row(x, t), save_v=row(traj_v, t), save_src=row(traj_s, t), save_aux=row(traj_aux, t),
                            save_acc=row(traj_acc, t), step0=t, **kw)
                net._gather(node, pop.row0, rows)
        ctx.node, ctx.net, ctx.steps, ctx.names = node, net, steps, names
        ctx.x = None if x is None else x.detach()
        ctx.saved = (traj_v, traj_s, traj_aux, traj_acc, traj_s_full)
        node._last_window = (traj_v, traj_s, traj_aux, None)
        out = traj_s if node._out_code == _abi.RP_OUT_SOURCE else traj_v
        return out.clone(), node._pack(pop.v, pop.source_local, pop.aux)

    @staticmethod
    def backward(ctx, g_traj, g_yend):
        node, net, T = ctx.node, ctx.net, ctx.steps
        pop = node._pop
        rows, n, dev = pop.rows, pop.n, pop.device
        traj_v, traj_s, traj_aux, traj_acc, traj_s_full = ctx.saved
        f32 = dict(dtype=torch.float32, device=dev)
        tpl = node._template

        def piece(name):
            if g_yend is None or name is None:
                return torch.zeros(rows, **f32)
            lo, hi = node._var_map[name]
            return g_yend.detach().to(**f32)[lo:hi].clone().contiguous()

        lam_v, lam_s = piece(tpl.reset_var), piece(tpl.source_var)
        lam_x = piece(node._aux_name) if node._aux_name else None
        g_out = None if g_traj is None else g_traj.detach().to(**f32).contiguous()
        need = ctx.needs_input_grad  # (node, net, steps, x, y_in, W, *params)
        grads = {name: torch.zeros(rows, **f32) for i, name in enumerate(ctx.names) if need[6 + i]}
        u_traj = torch.empty(T, rows, **f32)
        g_x = torch.empty(T, rows, **f32) if need[3] else None
        keep = pop.backward(traj_v, traj_s, g_out, node._out_code, lam_v, u_traj, traj_acc=traj_acc, grad_x=g_x,
                            lam_s=lam_s, slope=node._spike_slope, x=ctx.x, x_target=node._x_target, traj_aux=traj_aux,
                            lam_x=lam_x, grad_params=grads, exchange=_Exchange(net))
        g_W = None
        if need[5]:
            g_W = torch.empty(rows, n, **f32)
            pop.backend.outer_store_rect(rows, n, T, float(pop.dt), u_traj.data_ptr(), rows, traj_s_full.data_ptr(), n,
                                         g_W.data_ptr(), n, dev)
        if g_x is not None and ctx.x is not None and ctx.x.shape[1] == 1:
            g_x = g_x.sum(dim=1, keepdim=True)
            dist.all_reduce(g_x, group=net.group)
        del keep
        g_y = node._pack(lam_v, lam_s, lam_x) if need[4] else None
        return (None, None, None, g_x, g_y, g_W) + tuple(grads.get(name) for name in ctx.names)


class ShardedNetwork(Network):
    """``Network`` whose diff-eq population is row-sharded over the ranks of a ``torch.distributed`` group
    (one process per GPU).  Every rank builds the SAME graph from the SAME global arrays; this class keeps the
    rank's rows of the recurrent matrix, of per-neuron vectors and of input-edge weights, and the rank's columns of
    read-out weights.  ``run`` then integrates with one all-gather of the source vector per Euler step and one
    all-reduce of the (k-dimensional) read-out per window.  Node outputs, recorded variables and ``state`` are the
    rank's slices; read-out node outputs are complete on every rank.  ``fit_bptt`` trains the rank's rows of the
    recurrent matrix, its per-neuron parameters and its columns of a ``gd`` read-out with the sharded adjoint
    (reduce-scatter of the transposed product, one per step).
    """

    def __init__(self, dt: float, device: str = "cuda:0", dtype: torch.dtype = torch.float32, backend=None, group=None,
                 exchange: Optional[str] = None):
        """``exchange``: None / "rccl" (RCCL collectives enqueued in-library; host loop for gloo groups) or "direct" (the
        stream-ordered push exchange, :class:`DirectExchange`: all-gather of the source slices forward, rank-ordered sum of
        the partials in the adjoint; one sharded population per network; also RECTIPY_AMD_EXCHANGE=direct)."""
        import os
        super().__init__(dt, device=device, dtype=dtype, backend=backend)
        self._exchange_mode = exchange or os.environ.get("RECTIPY_AMD_EXCHANGE", "rccl")
        if self._exchange_mode not in ("rccl", "direct"):
            raise ValueError(f"exchange must be 'rccl' or 'direct', got {self._exchange_mode!r}")
        self.direct = None
        if not dist.is_initialized():
            raise RuntimeError("ShardedNetwork needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._sharded = {}  # label -> (row0, rows, n)
        self.comm = None  # rp_comm for the in-library exchange, set once the first collective has run
        self._exchange_checked = False

    def add_diffeq_node(self, label, node, input_var, output_var, weights=None, **kwargs):
        if weights is None:
            raise ValueError("a sharded population needs its `weights` (the global matrix or this rank's row block)")
        n = int(weights.shape[1])
        row0, rows = shard_bounds(n, self.world, self.rank)
        if weights.shape[0] == n:
            weights = weights[row0:row0 + rows]
        elif weights.shape[0] != rows:
            raise RuntimeError(f"weights must be [{n}, {n}] or this rank's [{rows}, {n}] block, got {tuple(weights.shape)}")
        if "node_vars" in kwargs:
            kwargs["node_vars"] = {k: _slice_rows(v, row0, rows, n) for k, v in kwargs["node_vars"].items()}
        new = super().add_diffeq_node(label, node, input_var, output_var, weights=weights, shard=(row0, n), **kwargs)
        self._sharded[label] = (row0, rows, n)
        self._wrap(new, row0, rows)
        self._gather(new, row0, rows, clone=True)
        if self._exchange_mode == "direct":
            if self.direct is not None:
                raise NotImplementedError("the direct exchange serves ONE sharded population per network (its buffers "
                                          "belong to the population)")
            if not hasattr(new._pop.backend, "forward_sharded") or new._pop.device.type != "cuda":
                raise RuntimeError("the direct exchange runs on the HIP backend only")
            self.direct = DirectExchange(new._pop, self.group)
            self.comm = self.direct.comm
            self._exchange_checked = True
        elif self.comm is None and not self._exchange_checked:
            self.comm = rccl_comm(self.group, new._pop.device, new._pop.backend)
            check_inlib_exchange(new._pop, self, 3, self.group)  # one-time guard; drops to the host loop on disagreement
            self._exchange_checked = True
        return new

    def _gather(self, node: RateNet, row0: int, rows: int, clone: bool = False):
        full = node._pop.source
        mine = full[row0:row0 + rows]
        if clone or dist.get_backend(self.group) != "nccl":
            mine = mine.clone()  # in-place (aliased) all-gather is an NCCL/RCCL feature
        dist.all_gather_into_tensor(full, mine, group=self.group)

    def _wrap(self, node: RateNet, row0: int, rows: int):
        """Per-step exchange after every step the node takes, whichever entry point is used."""
        net = self
        base_window, base_forward, base_reset = node.run_window, node.forward, node.reset
        base_before = getattr(node, "_before_step", None)

        def before_step(self_):
            """A state write by the holder of `y` (every rank runs the same script, so every rank sees its own slice
            written) changes this rank's slice of the source vector: the other ranks need it before the next step."""
            synced = base_before()
            if synced:
                net._gather(self_, row0, rows, clone=True)
            return synced

        def run_window(self_, steps, x=None, w_in=None, **kw):
            step0 = kw.pop("step0", 0)
            idx = kw.pop("sample_idx", 0)
            traj = kw.pop("traj", None)
            if net.comm is not None:
                return base_window(steps, x=x, w_in=w_in, step0=step0, sample_idx=idx, traj=traj, comm=net.comm, **kw)
            for t in range(steps):
                idx = base_window(1, x=None if x is None else x[t:t + 1], w_in=w_in, step0=step0 + t, sample_idx=idx,
                                  traj=None if traj is None else traj[t:t + 1], **kw)
                net._gather(self_, row0, rows)
            return idx

        def forward(self_, x):
            out = base_forward(x)
            net._gather(self_, row0, rows)
            return out

        def reset(self_, y=None, idx=None):
            base_reset(y=y, idx=idx)
            net._gather(self_, row0, rows, clone=True)

        def grad_window(self_, steps, x):
            pop = self_._pop
            if pop.spiking:
                from .spiking_autograd import TRAINABLE
                if pop.model not in TRAINABLE or pop.aux2 is not None or pop.model == "iku":
                    raise NotImplementedError(f"the '{pop.model}' template (population means) cannot be row-sharded")
                y_in = self_._y_grad
                if y_in is None:
                    y_in = self_.current_state.detach().requires_grad_(self_._y_requires_grad)
                params = [pop.params[q] for q in TRAINABLE[pop.model]]
                traj, y_end = _ShardedSpikeWindow.apply(self_, net, steps, x, y_in, pop.W, *params)
                self_._y_grad = y_end
                tv, ts, ta, _ = self_._last_window
                self_._y_prev = self_._pack(tv[-1], ts[-1], None if ta is None else ta[-1])
                return traj
            self_._before_step()
            v_in = self_._y_grad
            if v_in is None:
                v_in = pop.v.detach().clone().requires_grad_(self_._y_requires_grad)
            act = [pop.params[q] for q in ("r_max", "s", "v0")] if pop.model == "li_sigmoid" else []
            traj, v_end = _ShardedRateWindow.apply(self_, net, steps, x, v_in, pop.W, pop.params["tau"], pop.params["k"],
                                                   pop.params["eta"], *act)
            self_._y_grad = v_end
            return traj

        def forward_any(self_, x):  # per-step entry point: autograd window of one step when a gradient is wanted
            x2 = self_._as_input(x)
            if self_._wants_grad(x2):
                return grad_window(self_, 1, x2)[0]
            return forward(self_, x)

        if base_before is not None and not node._pop.spiking:
            node._before_step = types.MethodType(before_step, node)
        node.run_window = types.MethodType(run_window, node)
        node.forward = types.MethodType(forward_any, node)
        node.reset = types.MethodType(reset, node)
        node._grad_window = types.MethodType(grad_window, node)

    def add_edge(self, source, target, weights=None, train=None, edge_attrs=None, **kwargs):
        if train is not None and source not in self._sharded and target not in self._sharded:
            # a replicated trained edge sees the same gradient on every rank only downstream of a read-out; upstream of
            # a shard each rank would hold a partial gradient (its rows' share) and the sum is not implemented
            import networkx as nx
            if not any(nx.has_path(self.graph, lab, source) for lab in self._sharded):
                raise NotImplementedError("trained edges between replicated nodes upstream of a sharded population "
                                          "are not supported (their gradient would need an all-reduce)")
        if train not in (None, "gd") and source in self._sharded:
            raise NotImplementedError("only 'gd' training is available for read-outs of a sharded population")
        if weights is None and (source in self._sharded or target in self._sharded):
            raise ValueError("edges of a sharded population need explicit global weights (every rank must hold the same)")
        if target in self._sharded:      # input edge: this rank's rows
            row0, rows, n = self._sharded[target]
            w = weights if isinstance(weights, torch.Tensor) else np.asarray(weights)
            weights = w[row0:row0 + rows] if w.shape[0] == n else w.T[row0:row0 + rows]
        edge = None
        if source in self._sharded:      # read-out edge: this rank's columns, partial sums all-reduced
            row0, rows, n = self._sharded[source]
            w = weights if isinstance(weights, torch.Tensor) else np.asarray(weights)
            weights = w[:, row0:row0 + rows] if w.shape[1] == n else w.T[:, row0:row0 + rows]
            edge = super().add_edge(source, target, weights=weights, train=train, edge_attrs=edge_attrs, **kwargs)
            plain, group = edge.forward, self.group

            def forward(self_, x, **kw):
                return _AllReduceSum.apply(plain(x, **kw), group)

            edge.forward = types.MethodType(forward, edge)
            return edge
        return super().add_edge(source, target, weights=weights, train=train, edge_attrs=edge_attrs, **kwargs)

    def fit_rls(self, *a, **k):
        raise NotImplementedError("RLS / ridge read-out training is not available on a sharded population")

    fit_ridge = fit_rls
