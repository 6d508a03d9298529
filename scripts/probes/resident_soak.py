"""Soak of the resident launches: random sizes (up to 1536 on the 16-byte path, 1024 on the 4-byte path), models, step counts and windows for a fixed wall time; every
call is repeated with one launch per step from the same state and compared bit for bit; rp_async_status at the end.

    python scripts/probes/resident_soak.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectipy_amd import _abi  # noqa: E402
from rectipy_amd.engine import Population, default_backend  # noqa: E402

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def make(model, n, seed):
    gen = torch.Generator().manual_seed(seed)
    u = lambda lo, hi: torch.rand(n, generator=gen) * (hi - lo) + lo
    if model == "li_tanh":
        return Population(model, torch.randn(n, n, generator=gen) * 1.3 / n ** 0.5,
                          dict(tau=u(0.5, 2.0), k=u(0.5, 1.5), eta=torch.randn(n, generator=gen) * 0.1), 1e-2, dev,
                          v0=torch.randn(n, generator=gen) * 0.5)
    W = torch.rand(n, n, generator=gen) * 2.0 / n
    return Population("qif", W, dict(tau=u(0.8, 1.2), k=u(0.5, 1.5), eta=u(0.5, 3.0), tau_s=u(0.5, 2.0)), 5e-3, dev,
                      v0=torch.randn(n, generator=gen), theta=1.0, v_reset=-1.0)


t_end = time.time() + budget
calls = steps_total = 0
while time.time() < t_end:
    n = int(rng.integers(1, 385)) * 4 if rng.integers(0, 3) else int(rng.integers(1, 1025))   # (a third: any n -- the 4-byte path)
    model = ["li_tanh", "qif"][int(rng.integers(0, 2))]
    seed = int(rng.integers(0, 1 << 30))
    pops = []
    for resident in ("1", "0"):
        os.environ["RECTIPY_AMD_RESIDENT"] = resident
        pops.append(make(model, n, seed))
    for _ in range(int(rng.integers(1, 6))):   # a few consecutive windows on the same population
        T = int(rng.integers(4, 120))
        x = (torch.randn(T, n, generator=torch.Generator().manual_seed(seed + T)) * 0.5 + (1.5 if model == "qif" else 0.0)).to(dev)
        outs = []
        for resident, pop in zip(("1", "0"), pops):
            os.environ["RECTIPY_AMD_RESIDENT"] = resident
            traj = torch.zeros(T, n, device=dev)
            pop.forward(T, x=x, traj=traj, out_var=_abi.RP_OUT_SOURCE)
            outs.append((traj, pop.v.clone(), pop.source.clone()))
        for a, b in zip(outs[0], outs[1]):
            if not torch.equal(a, b):
                print(f"MISMATCH n={n} model={model} T={T}", flush=True)
                sys.exit(1)
        calls += 1
        steps_total += T
    if calls % 200 == 0:
        print(f"{calls} calls, {steps_total} steps", flush=True)
torch.cuda.synchronize()
default_backend().async_status()
print(f"resident soak: {calls} calls, {steps_total} steps, all bit-identical to a launch per step; no timeout reported", flush=True)
