"""Robustness checks on the GPU box: (1) 200 000 step launches enqueued by ONE rp_forward call; (2) repeated
fit_bptt calls do not grow device memory; (3) interleaving run / fit / reset keeps results reproducible."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd import Network  # noqa: E402
from rectipy_amd.engine import Population  # noqa: E402

dev = torch.device("cuda:0")
res = {}
N, T = 1024, 200000
W = torch.randn(N, N, device=dev) * (0.9 / N ** 0.5)
pop = Population("li_tanh", W, dict(tau=1.0, k=1.0, eta=0.0), 1e-2, dev)
x = torch.sin(torch.arange(T, device=dev) * 0.01).reshape(-1, 1).contiguous()
t0 = time.perf_counter()
pop.forward(T, x=x)
torch.cuda.synchronize()
res["long_call_steps_per_s"] = round(T / (time.perf_counter() - t0), 1)
res["long_call_finite"] = bool(torch.isfinite(pop.v).all())

rng = np.random.default_rng(0)
N = 4096
Wn = (rng.standard_normal((N, N)) * 1.2 / N ** 0.5).astype(np.float32)
net = Network(1e-2, device="cuda:0")
net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", input_var="li_op/I_ext",
                    output_var="tanh_op/r", weights=Wn, source_var="tanh_op/r", target_var="li_op/r_in",
                    dtype=torch.float32, train_params=["weights"])
net.add_func_node("out", 2, "identity")
net.add_edge("rnn", "out", weights=(rng.standard_normal((2, N)) / N ** 0.5).astype(np.float32), train="gd")
xs = rng.standard_normal((400, 1)).astype(np.float32)
tg = (rng.standard_normal((400, 2)) * 0.1).astype(np.float32)
mem = []
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    for i in range(12):
        net.fit_bptt(xs, tg, optimizer="adam", lr=1e-4, update_steps=100, sampling_steps=10, verbose=False)
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated() // (1 << 20))
res["memory_MiB_after_each_fit"] = mem
res["memory_stable"] = max(mem[3:]) - min(mem[3:]) <= 2

y0 = {k: v.clone() for k, v in net.state.items()}
a = net.run(xs, sampling_steps=10, verbose=False, enable_grad=False).to_numpy("out")
net.reset(y0)
b = net.run(xs, sampling_steps=10, verbose=False, enable_grad=False).to_numpy("out")
res["run_reset_run_identical"] = bool(np.array_equal(a, b))
print(json.dumps(res))
