"""The FeedbackNetwork leg of bench.py's extra.stepwise on its own (for rocprofv3: no other leg's kernels in the trace) with
wall-clock and HIP-event times side by side."""
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
    for N, Ts in ((4096, (1000, 4000)), (32768, (200, 1000))):
        h = N // 2
        kw = lambda W: dict(input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r",
                            target_var="li_op/r_in", dtype=torch.float32)
        fb = FeedbackNetwork(dt=1e-2, device=str(dev))
        fb.add_diffeq_node("a", tanh, **kw(bench.gen_rows(h, 0, h, dev)))
        fb.add_diffeq_node("b", tanh, **kw(bench.gen_rows(h, 0, h, dev)))
        fb.add_edge("a", "b", weights=bench.gen_rows(h, 0, h, dev, g=0.5))
        fb.add_edge("b", "a", weights=bench.gen_rows(h, 0, h, dev, g=0.5), feedback=True)
        gen = torch.Generator(device=dev).manual_seed(N)
        x = torch.randn(max(Ts), h, device=dev, generator=gen) * 0.1
        for T in Ts:
            fb.run(x[:max(2, T // 10)], sampling_steps=T, verbose=False, enable_grad=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            fb.run(x[:T], sampling_steps=T, verbose=False, enable_grad=False)
            e1.record()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            out[f"N{N}_T{T}"] = dict(wall_us_per_step=round((t2 - t0) / T * 1e6, 2), events_us_per_step=round(e0.elapsed_time(e1) / T * 1e3, 2),
                                     host_returned_after_ms=round((t1 - t0) * 1e3, 2))
        del fb, x
        torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
