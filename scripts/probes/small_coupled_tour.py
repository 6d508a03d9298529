"""us per step of FeedbackNetwork.run / fit_bptt on SMALL circuits (two tanh populations of h neurons, a -> b with feedback b -> a,
read-out of b): wall clock and host time (a gap = the host ran ahead of the device; none = the host or a hidden synchronisation
bounds the call), next to a single population of 2 h neurons through Network."""
import contextlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from rectipy_amd import FeedbackNetwork, Network  # noqa: E402

dev = torch.device("cuda:0")
tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
out = {}
T, win = 2000, 100


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return [round((time.perf_counter() - t0) / T * 1e6, 2), round(host / T * 1e6, 2)]


with contextlib.redirect_stdout(sys.stderr):
    for h in (128, 256, 512, 1024):
        gen = torch.Generator(device=dev).manual_seed(h)
        x = torch.randn(T, h, device=dev, generator=gen) * 0.1
        y = torch.randn(T, 3, device=dev, generator=gen) * 0.1
        for train in (None, "gd"):
            kw = lambda W: dict(input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r",
                                target_var="li_op/r_in", dtype=torch.float32, train_params=["weights"] if train else None)
            fb = FeedbackNetwork(dt=1e-2, device=str(dev))
            fb.add_diffeq_node("a", tanh, **kw(bench.gen_rows(h, 0, h, dev, g=0.8)))
            fb.add_diffeq_node("b", tanh, **kw(bench.gen_rows(h, 0, h, dev, g=0.8)))
            fb.add_func_node("out", 3, "identity")
            fb.add_edge("a", "b", weights=bench.gen_rows(h, 0, h, dev, g=0.5), train=train)
            fb.add_edge("b", "a", weights=bench.gen_rows(h, 0, h, dev, g=0.5), feedback=True, train=train)
            fb.add_edge("b", "out", weights=torch.randn(3, h, device=dev) / h ** 0.5, train="gd")
            if train is None:
                out[f"2x{h} run [wall, host]"] = timed(lambda: fb.run(x, sampling_steps=10, verbose=False, enable_grad=False))
                out[f"2x{h} fit_bptt read-out only"] = timed(lambda: fb.fit_bptt(x, y, optimizer="sgd", lr=1e-4, update_steps=win,
                                                                                  sampling_steps=win, verbose=False))
            else:
                out[f"2x{h} fit_bptt all matrices"] = timed(lambda: fb.fit_bptt(x, y, optimizer="sgd", lr=1e-4, update_steps=win,
                                                                                 sampling_steps=win, verbose=False))
                out[f"2x{h} fit_bptt all matrices, adam"] = timed(lambda: fb.fit_bptt(x, y, optimizer="adam", lr=1e-4, update_steps=win,
                                                                                       sampling_steps=win, verbose=False))
            del fb
        net = Network(dt=1e-2, device=str(dev))
        net.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="li_op/v", weights=bench.gen_rows(2 * h, 0, 2 * h, dev, g=0.8),
                            source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32, train_params=["weights"])
        net.add_func_node("out", 3, "identity")
        net.add_edge("rnn", "out", weights=torch.randn(3, 2 * h, device=dev) / h ** 0.5, train="gd")
        x2 = torch.randn(T, 2 * h, device=dev, generator=gen) * 0.1
        out[f"one population of {2 * h}: run"] = timed(lambda: net.run(x2, sampling_steps=10, verbose=False, enable_grad=False))
        out[f"one population of {2 * h}: fit_bptt"] = timed(lambda: net.fit_bptt(x2, y, optimizer="sgd", lr=1e-4, update_steps=win,
                                                                                  sampling_steps=win, verbose=False))
        del net
        print(out, flush=True)
print(json.dumps(out, indent=0))
