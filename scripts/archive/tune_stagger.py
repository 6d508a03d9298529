"""Experiment: rotate the start column chunk per 8-row group (decorrelates the lock-step march of all waves
through addresses that are equal modulo the 128 KiB row stride)."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import sys

import torch

from rectipy_amd import _lib
from rectipy_amd.engine import Population

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
STEPS = 40
dev = torch.device("cuda:0")
lib = _lib.load()
W = torch.randn(N, N, device=dev) * (1.5 / N ** 0.5)
pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
del W
x = torch.zeros(STEPS, 1, device=dev)


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


for rounds in range(2):
    for R, U in ((4, 2), (2, 4), (2, 2), (4, 4), (1, 8)):
        for stagger in (0, 1, 3, 7, 13, 31, 37):
            lib.rp_set_tuning(R, U, 1 | (stagger << 8), 0)
            pop.forward(3, x=x)
            best, med = timed(lambda: pop.forward(STEPS, x=x))
            print(json.dumps(dict(R=R, U=U, stagger=stagger, best_GBps=4.0 * N * N / (best / STEPS) / 1e6,
                                  med_GBps=4.0 * N * N / (med / STEPS) / 1e6)), flush=True)
lib.rp_set_tuning(0, 0, 1, 0)
