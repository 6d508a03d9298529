"""Sweep (unroll, strip width) of the adjoint column-strip kernel at N=32768 (GPU box)."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import sys

import torch

from rectipy_amd import _abi, _lib
from rectipy_amd.engine import Population

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = 20
dev = torch.device("cuda:0")
lib = _lib.load()
W = torch.randn(N, N, device=dev) * (1.5 / N ** 0.5)
pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
del W
tv, tr, ut = (torch.zeros(T, N, device=dev) for _ in range(3))
pop.forward(T, save_v=tv, save_src=tr)
g = torch.randn(T, N, device=dev) * 0.01


def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[0], ts[len(ts) // 2]


def run():
    lam = torch.zeros(N, device=dev)
    return pop.backward(tv, tr, g, _abi.RP_OUT_STATE0, lam, ut)


for rounds in range(2):
    for U, SW in ((8, 1), (16, 1), (32, 1), (12, 1), (4, 2), (2, 2), (3, 2), (5, 2), (6, 2), (8, 2)):
        lib.rp_set_bwd_tuning(U, SW)
        run()
        best, med = timed(run)
        print(json.dumps(dict(U=U, SW=SW, best_GBps=4.0 * N * N / (best / T) / 1e6, med_GBps=4.0 * N * N / (med / T) / 1e6)),
              flush=True)
lib.rp_set_bwd_tuning(0, 0)
