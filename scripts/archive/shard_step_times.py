"""Per-step kernel time of rank 0's row shard of the N = 32768 network for P = 1, 2, 4, 8 on ONE GPU (no exchange: the
source vector is left stale, the kernel's work and traffic are identical), with the streaming configuration the library
measures for each shape (rp_tune_forward).  Input to the efficiency model of DESIGN.md section 7."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json

import torch

from rectipy_amd.engine import Population

dev = torch.device("cuda:0")
N, STEPS = 32768, 200
x = torch.zeros(STEPS, 1, device=dev)
for P in (1, 2, 4, 8):
    rows = N // P
    W = torch.empty(rows, N, device=dev).normal_(0.0, 1.5 / N ** 0.5)
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev, n=N, row0=0)
    del W
    pop.source.copy_(torch.tanh(torch.randn(N, device=dev)))
    pop.forward(20, x=x)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pop.forward(STEPS, x=x)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / STEPS * 1e3)
    ts.sort()
    print(json.dumps(dict(P=P, rows=rows, n=N, us_per_step_median=round(ts[2], 2), us_min=round(ts[0], 2),
                          TBps=round(4.0 * rows * N / ts[2] / 1e6, 3), streaming=pop.backend.forward_tuning(pop._net()))),
          flush=True)
    del pop
    torch.cuda.empty_cache()
