"""Where does the first Network.run of a process spend its ~110 ms?  cProfile of the first and of the second 5-step call at
N = 32768 (the bench's extra.first_call), plus wall-clock stamps around the library call itself."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from rectipy_amd import Network  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
net = Network(dt=1e-2, device=str(dev))
net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", input_var="li_op/I_ext", output_var="li_op/v",
                    weights=bench.gen_rows(N, 0, N, dev), source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32)
x = torch.zeros(100, 1, device=dev)
torch.cuda.synchronize()
pop = net.get_node("rnn")._pop
inner = pop.backend.forward
stamps = []


def timed_forward(*a, **kw):
    t0 = time.perf_counter()
    inner(*a, **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    stamps.append((t1 - t0, time.perf_counter() - t1))


pop.backend.forward = timed_forward
for label in ("first", "second"):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    net.run(x[:5], sampling_steps=100, verbose=False, enable_grad=False)
    torch.cuda.synchronize()
    pr.disable()
    print(f"{label} call: {(time.perf_counter() - t0) * 1e3:.2f} ms; library call returned after {stamps[-1][0] * 1e3:.2f} ms, "
          f"GPU done {stamps[-1][1] * 1e3:.2f} ms later", flush=True)
    if label == "first":
        pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
