"""PCIe-inclusive rate of the headline workload (GPU box): the front end is handed HOST arrays (numpy W 4.3 GB, numpy
inputs [T, 1]), as the reference's users do; time = upload + 10^4 fused steps + read-back of the samples."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd import Network  # noqa: E402

N, T = 32768, 10000
rng = np.random.default_rng(0)
W = (rng.standard_normal((N, N), dtype=np.float32) * (1.5 / N ** 0.5)).astype(np.float32)
x = np.sin(2 * np.pi * 0.2 * np.arange(T) * 1e-2).astype(np.float32).reshape(-1, 1)
torch.zeros(1, device="cuda:0")  # context up before the clock starts
torch.cuda.synchronize()

t0 = time.perf_counter()
net = Network(dt=1e-2, device="cuda:0")
net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", input_var="li_op/I_ext",
                    output_var="li_op/v", weights=W, source_var="tanh_op/r", target_var="li_op/r_in",
                    dtype=torch.float32, node_vars={"all/li_op/tau": 10.0, "all/li_op/k": 1.0, "all/li_op/eta": 0.0})
torch.cuda.synchronize()
t1 = time.perf_counter()
obs = net.run(x, sampling_steps=100, verbose=False, enable_grad=False)
out = obs.to_numpy("out")
t2 = time.perf_counter()
print(json.dumps({"N": N, "steps": T, "upload_and_build_s": round(t1 - t0, 3), "run_and_readback_s": round(t2 - t1, 3),
                  "steps_per_s_resident": round(T / (t2 - t1), 1), "steps_per_s_pcie_inclusive": round(T / (t2 - t0), 1),
                  "W_upload_GBps": round(4.0 * N * N / (t1 - t0) / 1e9, 1), "samples": list(out.shape)}), flush=True)
