"""us per step of Network.fit_bptt (SGD, 100-step windows) at launch-bound sizes: resident launches against a launch per step."""
import contextlib
import io
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectipy_amd import Network  # noqa: E402

dev = torch.device("cuda:0")
tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
out = {}
for N in [int(a) for a in sys.argv[1:]] or (256, 512, 1024):
    T = 4000
    gen = torch.Generator(device=dev).manual_seed(0)
    W = torch.randn(N, N, device=dev, generator=gen) * 1.5 / N ** 0.5
    x = torch.randn(T, N, device=dev, generator=gen) * 0.5
    target = torch.randn(T, 3, device=dev, generator=gen) * 0.1
    for resident in ("1", "0"):
        os.environ["RECTIPY_AMD_RESIDENT"] = resident
        with contextlib.redirect_stdout(io.StringIO()):
            fit = Network(dt=1e-2, device=str(dev))
            fit.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="li_op/v", weights=W * 0.5, source_var="tanh_op/r",
                                target_var="li_op/r_in", dtype=torch.float32, train_params=["weights"])
            fit.add_func_node("out", 3, "identity")
            fit.add_edge("rnn", "out", weights=torch.randn(3, N, device=dev, generator=gen) / N ** 0.5, train="gd")
            fit.fit_bptt(x[:200], target[:200], optimizer="sgd", lr=1e-3, update_steps=100, sampling_steps=100, verbose=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fit.fit_bptt(x, target, optimizer="sgd", lr=1e-3, update_steps=100, sampling_steps=100, verbose=False)
            torch.cuda.synchronize()
        out[f"N{N}_{'resident' if resident == '1' else 'per_step'}"] = round((time.perf_counter() - t0) / T * 1e6, 3)
        print(out, file=sys.stderr, flush=True)
print(json.dumps(out))
