"""us per Euler step of Network.run at launch-bound sizes: the resident launch against a launch per step (RECTIPY_AMD_RESIDENT)."""
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
for N in [int(a) for a in sys.argv[1:]] or (100, 250, 512, 1000, 1024):
    T = 4000
    gen = torch.Generator(device=dev).manual_seed(0)
    W = torch.randn(N, N, device=dev, generator=gen) * 1.5 / N ** 0.5
    x = torch.randn(T, N, device=dev, generator=gen) * 0.5
    for resident in ("1", "0"):
        os.environ["RECTIPY_AMD_RESIDENT"] = resident
        with contextlib.redirect_stdout(io.StringIO()):
            net = Network(dt=1e-2, device=str(dev))
            net.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r",
                                target_var="li_op/r_in", dtype=torch.float32)
            net.run(x[:200], sampling_steps=1, verbose=False, enable_grad=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            net.run(x, sampling_steps=1, verbose=False, enable_grad=False)
            torch.cuda.synchronize()
        out[f"N{N}_{'resident' if resident == '1' else 'per_step'}"] = round((time.perf_counter() - t0) / T * 1e6, 3)
        print(out, file=sys.stderr, flush=True)
print(json.dumps(out))
