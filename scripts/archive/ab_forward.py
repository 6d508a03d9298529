"""A/B of a few streaming configurations of the tanh step kernel, interleaved and repeated (GPU box only).
usage: ab_forward.py N "R,U,mb;R,U,mb;..." [reps] [steps]"""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json

import torch

from rectipy_amd import _lib
from rectipy_amd.engine import Population

N = int(sys.argv[1])
cfgs = [tuple(int(v) for v in c.split(",")) for c in sys.argv[2].split(";")]
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = torch.device("cuda:0")
lib = _lib.load()
gen = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(N, N, device=dev, generator=gen) * (1.5 / N ** 0.5)
pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
del W
x = torch.sin(torch.arange(STEPS, device=dev) * 0.01).reshape(-1, 1).contiguous()
times = {c: [] for c in cfgs}
for rep in range(REPS):
    for c in cfgs:
        lib.rp_set_tuning(c[0], c[1], 1, c[2])
        pop.forward(3, x=x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pop.forward(STEPS, x=x)
        e1.record()
        torch.cuda.synchronize()
        times[c].append(e0.elapsed_time(e1) / STEPS)
lib.rp_set_tuning(0, 0, 1, 0)
for c in cfgs:
    t = sorted(times[c])
    print(json.dumps(dict(N=N, R=c[0], U=c[1], max_blocks=c[2], ms_min=round(t[0], 5), ms_med=round(t[len(t) // 2], 5),
                          ms_max=round(t[-1], 5), TBps_med=round(4.0 * N * N / t[len(t) // 2] / 1e9, 3))), flush=True)
