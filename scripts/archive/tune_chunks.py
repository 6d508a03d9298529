"""Sweep the adjoint's row-chunk count and strip shape at a given N (GPU box).  Benchmark-only: the chunk count is part
of the summation-order spec, production uses RC(n) of DESIGN.md."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd import _abi, _lib  # noqa: E402
from rectipy_amd.engine import Population  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
lib = _lib.load()
W = torch.randn(N, N, device=dev) * (1.5 / N ** 0.5)
pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
del W
tv, tr, ut = (torch.zeros(T, N, device=dev) for _ in range(3))
pop.forward(T, save_v=tv, save_src=tr)
g = torch.randn(T, N, device=dev) * 0.01


def run():
    lam = torch.zeros(N, device=dev)
    return pop.backward(tv, tr, g, _abi.RP_OUT_STATE0, lam, ut)


def timed(reps=5 if T <= 50 else 3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / T * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


# optional: argv[3] = "Q,Q,...", argv[4] = "U:SW,U:SW,...", argv[5] = "grid,grid,..." (0 = one workgroup per 4 items)
QS = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else ((16, 32, 48, 64) if T <= 50 else (16, 32, 64))
SHAPES = [tuple(int(w) for w in v.split(":")) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else \
    (((4, 2), (8, 1), (8, 2), (2, 2)) if T <= 50 else ((4, 2), (8, 1)))
GRIDS = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
for rounds in range(2 if len(sys.argv) <= 3 else 1):
    for Q in QS:
        for U, SW in SHAPES:
            for grid in GRIDS:
                lib.rp_set_bwd_chunks(Q)
                lib.rp_set_bwd_tuning(U, SW)
                lib.rp_set_bwd_grid(grid)
                run()
                best, med = timed()
                print(json.dumps(dict(N=N, Q=Q, U=U, SW=SW, grid=grid, best_us=round(best, 1), med_us=round(med, 1),
                                      med_GBps=round(4.0 * N * N / med / 1e3, 1))), flush=True)
lib.rp_set_bwd_chunks(0)
lib.rp_set_bwd_tuning(0, 0)
lib.rp_set_bwd_grid(0)
