"""FeedbackNetwork.fit_bptt at N = 32768 (two populations of 16384, both W and both edges trained, SGD windows of 50): the leg
of bench.py's extra.stepwise, alone, for a kernel trace.  Also times, without the profiler's help, the forward alone
(`run`), the forward WITH the window stores (the autograd node's forward, no backward), and the whole fit."""
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
N, T, win = 32768, 150, 50
h = N // 2
out = {}
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

    def timed(fn, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / steps * 1e6, 1)

    fit = lambda: fb.fit_bptt(x, y, optimizer="sgd", lr=1e-4, update_steps=win, sampling_steps=win, verbose=False)
    fb.fit_bptt(x[:win], y[:win], optimizer="sgd", lr=1e-4, update_steps=win, sampling_steps=win, verbose=False)
    out["fit_us_per_step"] = timed(fit, T)
    out["run_no_grad_us_per_step"] = timed(lambda: fb.run(x, sampling_steps=T, verbose=False, enable_grad=False), T)
    fb.compile()
    plan = fb._coupled_grad_plan()

    def fwd_with_stores():
        with torch.enable_grad():
            for a in range(0, T, win):
                fb._coupled_grad_outputs(plan, x[a:a + win], a, fit=True)
                fb.detach()
    out["forward_with_window_stores_us_per_step"] = timed(fwd_with_stores, T)
    out["run_no_grad_again_us_per_step"] = timed(lambda: fb.run(x, sampling_steps=T, verbose=False, enable_grad=False), T)
    out["fit_again_us_per_step"] = timed(fit, T)
    # plain SGD applies a trained matrix's update in the gradient GEMM in every window but the call's last complete one
    # (Network._fused_sgd_plan): a longer fit shows more of it; RECTIPY_AMD_FUSED_SGD=0 is the optimiser's passes everywhere
    x4, y4 = x.repeat(4, 1), y.repeat(4, 1)
    fit4 = lambda: fb.fit_bptt(x4, y4, optimizer="sgd", lr=1e-4, update_steps=win, sampling_steps=win, verbose=False)
    out["fit_12_windows_us_per_step"] = timed(fit4, 4 * T)
    os.environ["RECTIPY_AMD_FUSED_SGD"] = "0"
    out["fit_optimizer_passes_us_per_step"] = timed(fit, T)
    out["fit_12_windows_optimizer_passes_us_per_step"] = timed(fit4, 4 * T)
    del os.environ["RECTIPY_AMD_FUSED_SGD"]
print(json.dumps(out))
