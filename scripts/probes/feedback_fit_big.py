"""One N = 32768 FeedbackNetwork.fit_bptt run (two populations of 16384, all four matrices trained) for rocprofv3."""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from rectipy_amd import FeedbackNetwork  # noqa: E402

dev = torch.device("cuda:0")
tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
N, T, win = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 100, 50
h = N // 2
with contextlib.redirect_stdout(sys.stderr):
    kw = lambda W: dict(input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r",
                        target_var="li_op/r_in", dtype=torch.float32, train_params=["weights"])
    fb = FeedbackNetwork(dt=1e-2, device=str(dev))
    fb.add_diffeq_node("a", tanh, **kw(bench.gen_rows(h, 0, h, dev, g=0.8)))
    fb.add_diffeq_node("b", tanh, **kw(bench.gen_rows(h, 0, h, dev, g=0.8)))
    fb.add_func_node("out", 3, "identity")
    fb.add_edge("a", "b", weights=bench.gen_rows(h, 0, h, dev, g=0.5), train="gd")
    fb.add_edge("b", "a", weights=bench.gen_rows(h, 0, h, dev, g=0.5), feedback=True, train="gd")
    fb.add_edge("b", "out", weights=torch.randn(3, h, device=dev) / h ** 0.5, train="gd")
    gen = torch.Generator(device=dev).manual_seed(N)
    x = torch.randn(T, h, device=dev, generator=gen) * 0.1
    y = torch.randn(T, 3, device=dev, generator=gen) * 0.1
    fb.fit_bptt(x[:win], y[:win], optimizer="sgd", lr=1e-4, update_steps=win, sampling_steps=win, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fb.fit_bptt(x, y, optimizer="sgd", lr=1e-4, update_steps=win, sampling_steps=win, verbose=False)
    torch.cuda.synchronize()
print(round((time.perf_counter() - t0) / T * 1e6, 1), "us per step")
