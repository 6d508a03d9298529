"""Soak of the resident launch of coupled circuits (resident_coupled_kernel): random circuits of 2..4 small members (rate / QIF, state or
source outputs, an edge between every ordered pair that the slot budget allows), random step counts and consecutive windows, for a
fixed wall time; every call is repeated with one launch per step from the same state and compared bit for bit; rp_async_status at
the end.

    python scripts/probes/resident_coupled_soak.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rectipy_amd import _abi  # noqa: E402
from rectipy_amd.engine import default_backend, forward_coupled  # noqa: E402
from test_coupled import circuit, modes_for  # noqa: E402

dev = "cuda:0"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
calls = steps_total = 0
while time.time() < t_end:
    count = int(rng.integers(2, 5))
    cap = (1024 if count == 2 else 512) if rng.integers(0, 3) == 0 else (512 if count == 2 else 256)   # (a third: four rows per wave)
    sizes = tuple(int(rng.integers(1, cap // 4 + 1)) * 4 for _ in range(count))
    models = tuple(["li_tanh", "qif"][int(rng.integers(0, 2))] for _ in range(count))
    out_vars = tuple(int(rng.integers(0, 2)) for _ in range(count))
    seed = int(rng.integers(0, 1 << 30))
    # every ordered pair, thinned at random (a member may end without in-edges)
    keep = {(s_, t_) for s_ in range(count) for t_ in range(count) if s_ != t_ and rng.integers(0, 4)}
    sides = []
    for resident in ("1", "0"):
        os.environ["RECTIPY_AMD_RESIDENT"] = resident
        pops, edges = circuit(sizes, models, seed=seed, dev=dev, backend=None)
        hist = [[torch.zeros(p.rows, device=dev) for _ in range(3)] for p in pops]
        for m, p in enumerate(pops):
            hist[m][2].copy_(p.v if out_vars[m] == _abi.RP_OUT_STATE0 else p.source)
            p.reset_sampling()
        sides.append((pops, edges, hist, [0] * count))
    a = 0
    for _ in range(int(rng.integers(1, 5))):   # consecutive windows on the same circuits
        T = int(rng.integers(4, 90))
        x = (torch.randn(T, sizes[0], generator=torch.Generator().manual_seed(seed + a)) * 0.5).to(dev)
        outs = []
        for resident, (pops, edges, hist, idx) in zip(("1", "0"), sides):
            os.environ["RECTIPY_AMD_RESIDENT"] = resident
            trajs = [torch.zeros(T, p.rows, device=dev) for p in pops]
            samples = [torch.zeros(T, p.rows, device=dev) for p in pops]
            members = []
            for m, p in enumerate(pops):
                run = dict(steps=T, step0=a, out_var=out_vars[m], traj=trajs[m], cutoff=2, sampling_steps=3, samples=samples[m],
                           sample_idx=0, hist=hist[m])
                if m == 0:
                    run["x"] = x
                members.append(dict(pop=p, run=run, edges=[(edges[(s_, m)], s_, modes_for(models, s_, m)) for s_ in range(count)
                                                           if s_ != m and (s_, m) in keep]))
            forward_coupled(members)
            outs.append(trajs + samples + [p.v.clone() for p in pops] + [p.source.clone() for p in pops] + [h.clone() for hs in hist for h in hs])
        for q, (u, w) in enumerate(zip(outs[0], outs[1])):
            if not torch.equal(u, w):
                print(f"MISMATCH sizes={sizes} models={models} out_vars={out_vars} keep={sorted(keep)} T={T} a={a} tensor {q}", flush=True)
                sys.exit(1)
        a += T
        calls += 1
        steps_total += T
    if calls % 100 < 4:
        print(f"{calls} calls, {steps_total} steps", flush=True)
torch.cuda.synchronize()
default_backend().async_status()
print(f"resident_coupled_soak: {calls} calls, {steps_total} steps: bit-identical, no timeout", flush=True)
