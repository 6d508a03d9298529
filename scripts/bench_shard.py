"""BASELINE.json configs[4] per-GPU shape on ONE GPU: rank 0's shard of an N = 262144 network row-sharded over 8
GPUs (32768 rows x 262144 columns = 34.4 GB), stepped without the all-gather (the source vector is left stale, the
kernel's work and traffic are identical).  Also config 1 through the front end (N = 512, 2000 steps)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import time

import numpy as np
import torch

from rectipy_amd import Network
from rectipy_amd.engine import Population

dev = torch.device("cuda:0")
N, P = 262144, 8
rows = N // P
W = torch.empty(rows, N, device=dev)
gen = torch.Generator(device=dev).manual_seed(0)
for r in range(0, rows, 1024):
    W[r:r + 1024] = torch.randn(1024, N, device=dev, generator=gen)
W.mul_(1.5 / N ** 0.5)
pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev, n=N, row0=0)
pop.source.copy_(torch.tanh(torch.randn(N, device=dev)))
T = 200
x = torch.zeros(T, 1, device=dev)
pop.forward(10, x=x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
pop.forward(T, x=x)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / T
print(json.dumps({"workload": "shard 0 of 8 of N=262144 (32768 x 262144 fp32)", "ms_per_step": ms,
                  "steps_per_s_per_gpu_kernel_only": 1e3 / ms, "achieved_GBps": 4.0 * rows * N / (ms * 1e-3) / 1e9,
                  "frac_of_8TBps": 4.0 * rows * N / (ms * 1e-3) / 1e9 / 8000}))
del pop, W
torch.cuda.empty_cache()

# config 1: N = 512, 2000 steps, through Network.run (sampling_steps = 1)
n, T = 512, 2000
rng = np.random.default_rng(0)
net = Network(dt=1e-2, device="cuda:0")
net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", input_var="li_op/I_ext",
                    output_var="li_op/v", weights=rng.standard_normal((n, n)) * 1.5 / np.sqrt(n),
                    source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32)
inp = torch.tensor(rng.standard_normal((T, n)) * 0.5, dtype=torch.float32, device=dev)
net.run(inp[:100], verbose=False, enable_grad=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
obs = net.run(inp, sampling_steps=1, verbose=False, enable_grad=False)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(json.dumps({"workload": "config 1: tanh N=512, 2000 steps, Network.run, sampling_steps=1",
                  "steps_per_s": T / wall, "us_per_step": wall / T * 1e6, "samples": len(obs["steps"])}))
