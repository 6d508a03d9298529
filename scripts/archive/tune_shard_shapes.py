"""Step-kernel configuration for the row shards of the strong-scaling bench (rows = 32768 / P, n = 32768)."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json

import torch

from rectipy_amd import _lib
from rectipy_amd.engine import Population

dev = torch.device("cuda:0")
lib = _lib.load()
N = 32768
STEPS = 200
x = torch.zeros(STEPS, 1, device=dev)
for P in (2, 4, 8):
    rows = N // P
    W = torch.randn(rows, N, device=dev) * (1.5 / N ** 0.5)
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev, n=N, row0=0)
    pop.source.copy_(torch.tanh(torch.randn(N, device=dev)))
    for R, U in ((0, 0), (1, 2), (1, 4), (1, 8), (2, 2), (2, 4), (4, 1), (4, 2), (4, 4), (8, 2)):
        lib.rp_set_tuning(R, U, 1, 0)
        pop.forward(20, x=x)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pop.forward(STEPS, x=x)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / STEPS)
        print(json.dumps(dict(P=P, rows=rows, R=R, U=U, us_per_step=best * 1e3, GBps=4.0 * rows * N / best / 1e6)), flush=True)
    lib.rp_set_tuning(0, 0, 1, 0)
    del pop, W
