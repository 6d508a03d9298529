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


def shard_bounds(n: int, world: int, rank: int):
    if n % world:
        raise ValueError(f"n={n} must be divisible by the number of ranks ({world})")
    rows = n // world
    return rank * rows, rows


class ShardedPopulation:
    """One rank's shard plus the per-step exchange."""

    def __init__(self, model: str, weights_local, params_local: Dict[str, object], dt: float, device, n: int,
                 v0_local=None, s0_local=None, group=None, backend=None, **kw):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.row0, self.rows = shard_bounds(n, self.world, self.rank)
        self.pop = Population(model, weights_local, params_local, dt, device, v0=v0_local, s0=s0_local, n=n,
                              row0=self.row0, backend=backend, **kw)
        self.n = n
        self._gather_source()  # every rank needs the full initial source vector

    def _gather_source(self):
        full = self.pop.source
        dist.all_gather_into_tensor(full, full[self.row0:self.row0 + self.rows].clone(), group=self.group)

    def forward(self, steps: int, x: Optional[torch.Tensor] = None, out_var: int = _abi.RP_OUT_STATE0,
                traj: Optional[torch.Tensor] = None, step0: int = 0, **sampling):
        """`steps` Euler steps; `x` and `traj` are LOCAL ([steps, rows] or [steps, 1])."""
        idx = sampling.pop("sample_idx", 0)
        for t in range(steps):
            xt = None if x is None else x[t:t + 1]
            tr = None if traj is None else traj[t:t + 1]
            idx = self.pop.forward(1, x=xt, out_var=out_var, traj=tr, step0=step0 + t, sample_idx=idx, **sampling)
            full = self.pop.source  # the buffer the kernel just wrote this rank's slice into
            dist.all_gather_into_tensor(full, full[self.row0:self.row0 + self.rows], group=self.group)
        return idx


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


class ShardedNetwork(Network):
    """``Network`` whose diff-eq population is row-sharded over the ranks of a ``torch.distributed`` group
    (one process per GPU).  Every rank builds the SAME graph from the SAME global arrays; this class keeps the
    rank's rows of the recurrent matrix, of per-neuron vectors and of input-edge weights, and the rank's columns of
    read-out weights.  ``run`` then integrates with one all-gather of the source vector per Euler step and one
    all-reduce of the (k-dimensional) read-out per window.  Node outputs, recorded variables and ``state`` are the
    rank's slices; read-out node outputs are complete on every rank.  Forward integration only (``run`` / ``test``).
    """

    def __init__(self, dt: float, device: str = "cuda:0", dtype: torch.dtype = torch.float32, backend=None, group=None):
        super().__init__(dt, device=device, dtype=dtype, backend=backend)
        if not dist.is_initialized():
            raise RuntimeError("ShardedNetwork needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._sharded = {}  # label -> (row0, rows, n)

    def add_diffeq_node(self, label, node, input_var, output_var, weights=None, **kwargs):
        if weights is None:
            raise ValueError("a sharded population needs its `weights` (the global matrix or this rank's row block)")
        n = int(weights.shape[1])
        row0, rows = shard_bounds(n, self.world, self.rank)
        if weights.shape[0] == n:
            weights = weights[row0:row0 + rows]
        elif weights.shape[0] != rows:
            raise RuntimeError(f"weights must be [{n}, {n}] or this rank's [{rows}, {n}] block, got {tuple(weights.shape)}")
        if kwargs.get("train_params"):
            raise NotImplementedError("the sharded adjoint (reduce-scatter of W^T u) is not built yet: forward only")
        if "node_vars" in kwargs:
            kwargs["node_vars"] = {k: _slice_rows(v, row0, rows, n) for k, v in kwargs["node_vars"].items()}
        new = super().add_diffeq_node(label, node, input_var, output_var, weights=weights, shard=(row0, n), **kwargs)
        self._sharded[label] = (row0, rows, n)
        self._wrap(new, row0, rows)
        self._gather(new, row0, rows, clone=True)
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

        def run_window(self_, steps, x=None, w_in=None, **kw):
            step0 = kw.pop("step0", 0)
            idx = kw.pop("sample_idx", 0)
            traj = kw.pop("traj", None)
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

        node.run_window = types.MethodType(run_window, node)
        node.forward = types.MethodType(forward, node)
        node.reset = types.MethodType(reset, node)

    def add_edge(self, source, target, weights=None, train=None, edge_attrs=None, **kwargs):
        if train is not None and (source in self._sharded or target in self._sharded):
            raise NotImplementedError("trained edges next to a sharded population are not supported yet")
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
            edge = super().add_edge(source, target, weights=weights, train=None, edge_attrs=edge_attrs, **kwargs)
            plain, group = edge.forward, self.group

            def forward(self_, x, **kw):
                part = plain(x, **kw).contiguous()
                dist.all_reduce(part, group=group)
                return part

            edge.forward = types.MethodType(forward, edge)
            return edge
        return super().add_edge(source, target, weights=weights, train=train, edge_attrs=edge_attrs, **kwargs)

    def fit_bptt(self, *a, **k):
        raise NotImplementedError("sharded training is not built yet (forward integration only)")

    fit_rls = fit_ridge = fit_bptt
