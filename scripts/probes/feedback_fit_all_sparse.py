"""FeedbackNetwork.fit_bptt, two populations of N/2 whose recurrent matrices AND edges are fixed with p = 0.1; trained: tau and
eta of both populations and the read-out.  us per step with everything streamed packed (forward: rp_net.sp_*, rp_run.edge_sp;
adjoint: rp_bwd.spt_*, rp_bwd_member.edge_spt) and densely (sparse_weights=False, no packed edges)."""
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
N, T, win, p = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 100, 50, 0.1
h = N // 2
gen = torch.Generator(device=dev).manual_seed(N)


def sparse(g):
    W = torch.randn(h, h, device=dev, generator=gen) * (g / (h * p) ** 0.5)
    return W * (torch.rand(h, h, device=dev, generator=gen) < p)


mats = [sparse(0.9), sparse(0.9), sparse(0.5), sparse(0.5)]
x = torch.randn(T, h, device=dev, generator=gen) * 0.1
y = torch.randn(T, 3, device=dev, generator=gen) * 0.1
out = {}
with contextlib.redirect_stdout(sys.stderr):
    for packed in (True, False):
        kw = lambda W: dict(input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r", target_var="li_op/r_in",
                            dtype=torch.float32, train_params=["li_op/tau", "li_op/eta"], sparse_weights="auto" if packed else False)
        fb = FeedbackNetwork(dt=1e-2, device=str(dev))
        fb.add_diffeq_node("a", tanh, **kw(mats[0].clone()))
        fb.add_diffeq_node("b", tanh, **kw(mats[1].clone()))
        fb.add_func_node("out", 3, "identity")
        fb.add_edge("a", "b", weights=mats[2].clone())
        fb.add_edge("b", "a", weights=mats[3].clone(), feedback=True)
        fb.add_edge("b", "out", weights=torch.randn(3, h, device=dev) / h ** 0.5, train="gd")
        fb.fit_bptt(x[:win], y[:win], optimizer="sgd", lr=1e-4, update_steps=win, sampling_steps=win, verbose=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fb.fit_bptt(x, y, optimizer="sgd", lr=1e-4, update_steps=win, sampling_steps=win, verbose=False)
        torch.cuda.synchronize()
        out["packed" if packed else "dense"] = round((time.perf_counter() - t0) / T * 1e6, 1)
        del fb
        torch.cuda.empty_cache()
print(json.dumps(out))
