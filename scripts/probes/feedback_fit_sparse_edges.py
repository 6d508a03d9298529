"""FeedbackNetwork.fit_bptt, two populations of N/2 with TRAINED dense recurrent matrices and FIXED sparse (p = 0.1) edges
between them (a -> b, feedback b -> a): us per step with the edges streamed packed (forward: rp_run.edge_sp, adjoint:
rp_bwd_member.edge_spt) and densely (PACKED_EDGE_MAX_DENSITY = 0)."""
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
N, T, win, p = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 100, 50, 0.1
h = N // 2
gen = torch.Generator(device=dev).manual_seed(N)


def sparse_edge(g):
    W = torch.randn(h, h, device=dev, generator=gen) * (g / (h * p) ** 0.5)
    return W * (torch.rand(h, h, device=dev, generator=gen) < p)


E_ab, E_ba = sparse_edge(0.5), sparse_edge(0.5)
x = torch.randn(T, h, device=dev, generator=gen) * 0.1
y = torch.randn(T, 3, device=dev, generator=gen) * 0.1
out = {}
with contextlib.redirect_stdout(sys.stderr):
    for packed in (True, False):
        kw = lambda W: dict(input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r",
                            target_var="li_op/r_in", dtype=torch.float32, train_params=["weights"])
        fb = FeedbackNetwork(dt=1e-2, device=str(dev))
        if not packed:
            fb.PACKED_EDGE_MAX_DENSITY = 0.0
        fb.add_diffeq_node("a", tanh, **kw(bench.gen_rows(h, 0, h, dev, g=0.8)))
        fb.add_diffeq_node("b", tanh, **kw(bench.gen_rows(h, 0, h, dev, g=0.8)))
        fb.add_func_node("out", 3, "identity")
        fb.add_edge("a", "b", weights=E_ab.clone())
        fb.add_edge("b", "a", weights=E_ba.clone(), feedback=True)
        fb.add_edge("b", "out", weights=torch.randn(3, h, device=dev) / h ** 0.5, train="gd")
        fb.fit_bptt(x[:win], y[:win], optimizer="sgd", lr=1e-4, update_steps=win, sampling_steps=win, verbose=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fb.fit_bptt(x, y, optimizer="sgd", lr=1e-4, update_steps=win, sampling_steps=win, verbose=False)
        torch.cuda.synchronize()
        out["packed_edges" if packed else "dense_edges"] = round((time.perf_counter() - t0) / T * 1e6, 1)
        del fb
        torch.cuda.empty_cache()
print(json.dumps(out))
