"""Sweep the streaming configuration of the tanh step kernel at N=32768 (GPU box only)."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import json
import sys
import time

import torch

from rectipy_amd import _lib
from rectipy_amd.engine import Population

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
lib = _lib.load()
gen = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(N, N, device=dev, generator=gen) * (1.5 / N ** 0.5)
pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
del W
x = torch.sin(torch.arange(STEPS, device=dev) * 0.01).reshape(-1, 1).contiguous()


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


# raw HBM read probe over the same buffer
out = torch.zeros(4, device=dev)
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
ms = timed(lambda: lib.rp_stream_probe(pop.W.data_ptr(), pop.W.numel(), out.data_ptr(), stream))
print(json.dumps({"probe_GBps": pop.W.numel() * 4 / ms / 1e6}), flush=True)

results = []
cfgs = [(0, 0, 1, 0)]
for R in (1, 2, 4, 8):
    for U in (1, 2, 4, 8, 16):
        for nt in (1, 0):
            for mb in (0, 1024, 2048, 4096):
                cfgs.append((R, U, nt, mb))
valid = {(1, 2), (1, 4), (1, 8), (1, 16), (2, 8), (2, 1), (2, 2), (2, 4), (4, 1), (4, 2), (4, 4), (8, 1), (8, 2)}
for R, U, nt, mb in cfgs:
    if R and (R, U) not in valid:
        continue
    lib.rp_set_tuning(R, U, nt, mb)
    pop.forward(3, x=x)
    ms = timed(lambda: pop.forward(STEPS, x=x)) / STEPS
    gbps = 4.0 * N * N / ms / 1e6
    results.append(dict(R=R, U=U, nt=nt, max_blocks=mb, ms_per_step=ms, GBps=gbps))
    print(json.dumps(results[-1]), flush=True)
lib.rp_set_tuning(0, 0, 1, 0)
best = max(results, key=lambda r: r["GBps"])
print("BEST", json.dumps(best))
