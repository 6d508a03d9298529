"""BASELINE.json configs[3]: QIF spiking population, N = 65536 dense, spike/reset path (GPU box).
W = rand(N, N) * 20 / N, tau = k = tau_s = 1, eta = -5, v0 = -2, theta = 100, v_reset = -100, dt = 1e-3,
constant input 15 (SURVEY.md 8d "C4")."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import sys
import time

import torch

from rectipy_amd import _abi
from rectipy_amd.engine import Population

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
W = torch.empty(N, N, device=dev)
for r in range(0, N, 4096):
    W[r:r + 4096] = torch.rand(min(4096, N - r), N, device=dev, generator=gen)
W.mul_(20.0 / N)
pop = Population("qif", W, dict(tau=1.0, k=1.0, eta=-5.0, tau_s=1.0), 1e-3, dev, v0=-2.0, s0=0.0, theta=100.0,
                 v_reset=-100.0)
del W
x = torch.full((T, 1), 15.0, device=dev)
ev = torch.zeros(8_000_000, 2, dtype=torch.int32, device=dev)
cnt = torch.zeros(1, dtype=torch.int32, device=dev)
S = T // 100 + 1
samples = torch.zeros(S, N, device=dev)
pop.forward(100, x=x[:100])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
pop.forward(T, x=x, step0=100, out_var=_abi.RP_OUT_SOURCE, sampling_steps=100, samples=samples, spike_events=ev,
            spike_count=cnt)
e1.record()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
ms = e0.elapsed_time(e1) / T
algo = 4.0 * N * N + 4.0 * 12 * N
print(json.dumps({"workload": f"QIF N={N} dense, {T} steps, spike/reset", "steps_per_s": T / wall, "ms_per_step": ms,
                  "achieved_GBps": algo / (ms * 1e-3) / 1e9, "frac_of_8TBps": algo / (ms * 1e-3) / 1e9 / 8000.0,
                  "spikes": int(cnt.item()), "spikes_per_neuron_per_unit_time": int(cnt.item()) / N / (T * 1e-3),
                  "mean_s": float(samples[-1].mean()), "finite": bool(torch.isfinite(pop.v).all().item())}))
