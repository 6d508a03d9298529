"""Host-side cost per library call vs per in-library step at small N (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import time

import torch

from rectipy_amd.engine import Population

dev = torch.device("cuda:0")
for N in (512, 2048, 4096):
    W = torch.randn(N, N, device=dev) * (1.5 / N ** 0.5)
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
    T = 2000
    x = torch.randn(T, 1, device=dev)
    samples = torch.zeros(T, N, device=dev)
    pop.forward(50, x=x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pop.forward(T, x=x, samples=samples, sampling_steps=1)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_fused = time.perf_counter() - t0
    pop.reset_sampling()
    t0 = time.perf_counter()
    idx = 0
    for t in range(T):
        idx = pop.forward(1, x=x[t:t + 1], samples=samples, sampling_steps=1, step0=t, sample_idx=idx)
    t_issue1 = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_step = time.perf_counter() - t0
    print(json.dumps(dict(N=N, fused_us_per_step=t_fused / T * 1e6, fused_issue_us=t_issue / T * 1e6,
                          percall_us_per_step=t_step / T * 1e6, percall_issue_us=t_issue1 / T * 1e6,
                          fused_steps_per_s=T / t_fused)), flush=True)
