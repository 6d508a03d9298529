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
