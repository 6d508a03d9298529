"""GPU box: where do the 12 us per step of Network.fit_bptt at N = 512 go?  Wall clock per step for windows of 50 / 100 / 200 /
500 steps (the per-window work shows as the slope in 1 / window), host time of the python profile's top entries, and the
sum of kernel durations when run under rocprofv3 --kernel-trace (post-processed elsewhere).

    python scripts/probes/small_n_fit.py [N]
"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectipy_amd import Network  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = 2000
dev = torch.device("cuda:0")
tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
gen = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(N, N, device=dev, generator=gen) * 1.5 / N ** 0.5
x = torch.randn(T, N, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 0.5
target = torch.randn(T, 3, device=dev, generator=gen) * 0.1


def build():
    fit = Network(dt=1e-2, device=str(dev))
    with contextlib.redirect_stdout(io.StringIO()):
        fit.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="li_op/v", weights=W * 0.5, source_var="tanh_op/r",
                            target_var="li_op/r_in", dtype=torch.float32, train_params=["weights"])
    fit.add_func_node("out", 3, "identity")
    fit.add_edge("rnn", "out", weights=torch.randn(3, N, device=dev, generator=gen) / N ** 0.5, train="gd")
    return fit


for win in (50, 100, 200, 500):
    fit = build()
    fit.fit_bptt(x[:2 * win], target[:2 * win], optimizer="sgd", lr=1e-3, update_steps=win, sampling_steps=win, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit.fit_bptt(x, target, optimizer="sgd", lr=1e-3, update_steps=win, sampling_steps=win, verbose=False)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"window {win}: {t_all / T * 1e6:.2f} us per step (host returned after {t_host / T * 1e6:.2f}); per window {t_all / (T / win) * 1e6:.0f} us", flush=True)

fit = build()
fit.fit_bptt(x[:200], target[:200], optimizer="sgd", lr=1e-3, update_steps=100, sampling_steps=100, verbose=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
fit.fit_bptt(x, target, optimizer="sgd", lr=1e-3, update_steps=100, sampling_steps=100, verbose=False)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
