"""Front-end throughput at the reference's own scale (BASELINE.json configs[0]: N=512): Network.run and
Network.fit_bptt wall time per step, with a cProfile of the host side (GPU box)."""
import cProfile
import io
import json
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd import Network  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = 2000
US = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(0)
W = (rng.standard_normal((N, N)) * 1.5 / N ** 0.5).astype(np.float32)
x = (0.5 * rng.standard_normal((T, N))).astype(np.float32)
tgt = (0.1 * rng.standard_normal((T, 3))).astype(np.float32)
W_out = (rng.standard_normal((3, N)) / N ** 0.5).astype(np.float32)


def build(train):
    net = Network(dt=1e-2, device="cuda:0")
    net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", input_var="li_op/I_ext",
                        output_var="tanh_op/r", weights=W, source_var="tanh_op/r", target_var="li_op/r_in",
                        dtype=torch.float32, train_params=["weights"] if train else None)
    net.add_func_node("out", 3, "identity")
    net.add_edge("rnn", "out", weights=W_out, train="gd" if train else None)
    return net


res = {"N": N, "T": T, "update_steps": US}
net = build(False)
net.run(x, sampling_steps=1, verbose=False, enable_grad=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
net.run(x, sampling_steps=1, verbose=False, enable_grad=False)
torch.cuda.synchronize()
res["run_us_per_step"] = round((time.perf_counter() - t0) / T * 1e6, 2)

net = build(True)
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    net.fit_bptt(x, tgt, optimizer="adam", lr=1e-3, update_steps=US, sampling_steps=1, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net.fit_bptt(x, tgt, optimizer="adam", lr=1e-3, update_steps=US, sampling_steps=1, verbose=False)
    torch.cuda.synchronize()
    res["fit_bptt_us_per_step"] = round((time.perf_counter() - t0) / T * 1e6, 2)
    pr = cProfile.Profile()
    pr.enable()
    net.fit_bptt(x, tgt, optimizer="adam", lr=1e-3, update_steps=US, sampling_steps=1, verbose=False)
    torch.cuda.synchronize()
    pr.disable()
print(json.dumps(res), flush=True)
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(35)
print(buf.getvalue()[:6000])
