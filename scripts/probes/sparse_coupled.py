"""Two populations of 16384 tanh neurons coupled by dense edges, one a feedback edge, every matrix thinned to p = 0.1
(FeedbackNetwork.run): us per step with the matrices streamed packed (default) and densely (sparse_weights=False)."""
import contextlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rectipy_amd import FeedbackNetwork  # noqa: E402

dev = torch.device("cuda:0")
tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
h, T, p = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 300, 0.1
gen = torch.Generator(device=dev).manual_seed(1)


def mat(rows, cols, g):
    W = torch.randn(rows, cols, device=dev, generator=gen) * (g / (cols * p) ** 0.5)
    return W * (torch.rand(rows, cols, device=dev, generator=gen) < p)


mats = [mat(h, h, 1.0), mat(h, h, 1.0), mat(h, h, 0.5), mat(h, h, 0.5)]
x = torch.randn(T, h, device=dev, generator=gen) * 0.1
out = {}
with contextlib.redirect_stdout(sys.stderr):
    for mode in ("auto", False):
        fb = FeedbackNetwork(dt=1e-2, device=str(dev))
        kw = lambda W: dict(input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r",
                            target_var="li_op/r_in", dtype=torch.float32, sparse_weights=mode)
        fb.add_diffeq_node("a", tanh, **kw(mats[0].clone()))
        fb.add_diffeq_node("b", tanh, **kw(mats[1].clone()))
        fb.add_edge("a", "b", weights=mats[2].clone())
        fb.add_edge("b", "a", weights=mats[3].clone(), feedback=True)
        fb.run(x[:20], sampling_steps=20, verbose=False, enable_grad=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fb.run(x, sampling_steps=T, verbose=False, enable_grad=False)
        torch.cuda.synchronize()
        out["packed" if mode == "auto" else "dense"] = round((time.perf_counter() - t0) / T * 1e6, 1)
        out["packed_edges" if mode == "auto" else "dense_edges_packed"] = sum(
            getattr(fb.get_edge(s_, t_), "_packed_cache", (None, None))[1] is not None for s_, t_ in (("a", "b"), ("b", "a")))
        del fb
        torch.cuda.empty_cache()
print(json.dumps(out))
