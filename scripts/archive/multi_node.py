"""Per-step cost of a two-population feed-forward network (p1 -> p2), which has no single-population chain plan, vs a
single population of the same size (GPU box)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd import Network  # noqa: E402

TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
rng = np.random.default_rng(0)
res = {}
for N in (512, 4096):
    T = 2000
    W1, W2 = ((rng.standard_normal((N, N)) * 1.2 / N ** 0.5).astype(np.float32) for _ in range(2))
    J = (rng.standard_normal((N, N)) / N ** 0.5).astype(np.float32)
    x = rng.standard_normal((T, 1)).astype(np.float32)
    net = Network(1e-2, device="cuda:0")
    for lab, W in (("p1", W1), ("p2", W2)):
        net.add_diffeq_node(lab, TANH, input_var="li_op/I_ext", output_var="tanh_op/r", weights=W, source_var="tanh_op/r",
                            target_var="li_op/r_in", dtype=torch.float32)
    net.add_edge("p1", "p2", weights=J)
    net.run(x[:50], sampling_steps=10, verbose=False, enable_grad=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    obs = net.run(x, sampling_steps=10, verbose=False, enable_grad=False)
    torch.cuda.synchronize()
    res[f"two_nodes_N{N}_us_per_step"] = round((time.perf_counter() - t0) / T * 1e6, 1)
    res[f"out_checksum_N{N}"] = float(np.abs(obs.to_numpy("out")).sum())
print(json.dumps(res))
