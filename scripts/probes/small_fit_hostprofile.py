"""Where the host spends a fit_bptt window at N = 512 (the call is host-bound: `host` = `wall`): cProfile over 40 windows of 100 steps."""
import cProfile
import contextlib
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rectipy_amd import FeedbackNetwork, Network  # noqa: E402

dev = torch.device("cuda:0")
tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
N, T = 512, 4000
gen = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(N, N, device=dev, generator=gen) * 1.5 / N ** 0.5
x = torch.randn(T, N, device=dev, generator=gen) * 0.5
target = torch.randn(T, 3, device=dev, generator=gen) * 0.1
which = sys.argv[1] if len(sys.argv) > 1 else "single"
with contextlib.redirect_stdout(io.StringIO()):
    if which == "single":
        fit = Network(dt=1e-2, device=str(dev))
        fit.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="li_op/v", weights=W * 0.5, source_var="tanh_op/r",
                            target_var="li_op/r_in", dtype=torch.float32, train_params=["weights"])
        fit.add_func_node("out", 3, "identity")
        fit.add_edge("rnn", "out", weights=torch.randn(3, N, device=dev, generator=gen) / N ** 0.5, train="gd")
        xin = x
    else:
        h = N // 2
        fit = FeedbackNetwork(dt=1e-2, device=str(dev))
        for lab in "ab":
            fit.add_diffeq_node(lab, tanh, input_var="li_op/I_ext", output_var="li_op/v", weights=W[:h, :h].contiguous() * 0.5,
                                source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32, train_params=["weights"])
        fit.add_func_node("out", 3, "identity")
        fit.add_edge("a", "b", weights=W[:h, h:].contiguous() * 0.3, train="gd")
        fit.add_edge("b", "a", weights=W[h:, :h].contiguous() * 0.3, feedback=True, train="gd")
        fit.add_edge("b", "out", weights=torch.randn(3, h, device=dev, generator=gen) / h ** 0.5, train="gd")
        xin = x[:, :h].contiguous()
    call = lambda: fit.fit_bptt(xin, target, optimizer="sgd", lr=1e-3, update_steps=100, sampling_steps=100, verbose=False)
    call()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    call()
    torch.cuda.synchronize()
    pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue())

# the autograd node's backward runs on the engine's device thread (not under cProfile): time its parts by hand
import time  # noqa: E402

from rectipy_amd import nodes as _nodes  # noqa: E402
from rectipy_amd.engine import Population  # noqa: E402

acc = {}


def timed(name, fn):
    def inner(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return inner


if which == "single":
    Population.backward = timed("Population.backward (the adjoint sweep's library call)", Population.backward)
    Population.outer_product = timed("Population.outer_product (dW GEMM)", Population.outer_product)
    Population.outer_accum = timed("Population.outer_accum (dW GEMM into W: plain SGD)", Population.outer_accum)
    Population.forward = timed("Population.forward (the window's library call)", Population.forward)
    _nodes._RateWindow.backward = staticmethod(timed("_RateWindow.backward (whole)", _nodes._RateWindow.backward))
    _nodes._RateWindow.forward = staticmethod(timed("_RateWindow.forward (whole)", _nodes._RateWindow.forward))
    with contextlib.redirect_stdout(io.StringIO()):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    print(f"40 windows: host {host * 1e3:.2f} ms, wall {wall * 1e3:.2f} ms")
    for k_, v_ in sorted(acc.items(), key=lambda kv: -kv[1]):
        print(f"  {v_ * 1e3 / 40:8.3f} ms per window  {k_}")
