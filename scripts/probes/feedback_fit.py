"""fit_bptt through a FeedbackNetwork (two tanh populations of N/2, a -> b with feedback b -> a, read-out of b; W of both
populations and both edges trained): us per step, wall clock."""
import contextlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from rectipy_amd import FeedbackNetwork  # noqa: E402

dev = torch.device("cuda:0")
tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
out = {}
with contextlib.redirect_stdout(sys.stderr):
    for N, T, win in ((1024, 400, 100), (4096, 400, 100), (32768, 100, 50)):
        h = N // 2
        for trained_edges, walk in ((False, False), (True, False), (True, True)):
            if walk and N > 4096:
                continue
            kw = lambda W: dict(input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r",
                                target_var="li_op/r_in", dtype=torch.float32, train_params=["weights"])
            fb = FeedbackNetwork(dt=1e-2, device=str(dev))
            fb.add_diffeq_node("a", tanh, **kw(bench.gen_rows(h, 0, h, dev, g=0.8)))
            fb.add_diffeq_node("b", tanh, **kw(bench.gen_rows(h, 0, h, dev, g=0.8)))
            fb.add_func_node("out", 3, "identity")
            tr = "gd" if trained_edges else None
            fb.add_edge("a", "b", weights=bench.gen_rows(h, 0, h, dev, g=0.5), train=tr)
            fb.add_edge("b", "a", weights=bench.gen_rows(h, 0, h, dev, g=0.5), feedback=True, train=tr)
            fb.add_edge("b", "out", weights=torch.randn(3, h, device=dev) / h ** 0.5, train="gd")
            fb._force_stepwise_adjoint = walk  # True: autograd over the per-step graph walk (what r03 did)
            gen = torch.Generator(device=dev).manual_seed(N)
            x = torch.randn(T, h, device=dev, generator=gen) * 0.1
            y = torch.randn(T, 3, device=dev, generator=gen) * 0.1
            fb.fit_bptt(x[:win], y[:win], optimizer="sgd", lr=1e-4, update_steps=win, sampling_steps=win, verbose=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fb.fit_bptt(x, y, optimizer="sgd", lr=1e-4, update_steps=win, sampling_steps=win, verbose=False)
            torch.cuda.synchronize()
            out[f"N{N}_edges_{'trained' if trained_edges else 'fixed'}{'_walk' if walk else ''}"] = round((time.perf_counter() - t0) / T * 1e6, 1)
            print(out, flush=True)
            del fb
            torch.cuda.empty_cache()
print(json.dumps(out))
