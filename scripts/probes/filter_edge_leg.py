"""Two populations of N/2 joined by a LinearFilter edge (a -> b through y <- F y + x, feedback b -> a) through
FeedbackNetwork.run -- the leg `feedback_pair_with_LinearFilter_edge` of bench.py's extra.stepwise, alone, for a kernel trace
(what runs: the coupled step kernels and the filter's mat-vec kernel; no rocBLAS, no host copies of the edge's state)."""
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
    for N, T in ((4096, 1000), (32768, 200)):
        h = N // 2
        kw = lambda W: dict(input_var="li_op/I_ext", output_var="li_op/v", weights=W, source_var="tanh_op/r",
                            target_var="li_op/r_in", dtype=torch.float32)
        fb = FeedbackNetwork(dt=1e-2, device=str(dev))
        fb.add_diffeq_node("a", tanh, **kw(bench.gen_rows(h, 0, h, dev)))
        fb.add_diffeq_node("b", tanh, **kw(bench.gen_rows(h, 0, h, dev)))
        fb.add_edge("a", "b", weights=bench.gen_rows(h, 0, h, dev, g=0.5), filter_weights=bench.gen_rows(h, 0, h, dev, g=0.5))
        fb.add_edge("b", "a", weights=bench.gen_rows(h, 0, h, dev, g=0.5), feedback=True)
        x = torch.randn(T, h, device=dev, generator=torch.Generator(device=dev).manual_seed(N)) * 0.1
        fb.run(x[:T // 10], sampling_steps=T, verbose=False, enable_grad=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fb.run(x, sampling_steps=T, verbose=False, enable_grad=False)
        torch.cuda.synchronize()
        out[f"N{N}_us_per_step"] = round((time.perf_counter() - t0) / T * 1e6, 1)
        del fb
        torch.cuda.empty_cache()
print(json.dumps(out))
