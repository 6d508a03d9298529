"""Does the resident launch of a coupled circuit return to the host while EARLIER work of the stream is still running?  (Its members'
argument blocks travel by value for up to three members; the four-member form copies them from host memory.)  A long torch
operation is queued first; the host time of the following FeedbackNetwork.run call is compared with the operation's duration."""
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
    for count in (2, 4):
        h = 128
        fb = FeedbackNetwork(dt=1e-2, device=str(dev))
        labs = "abcd"[:count]
        for lab in labs:
            fb.add_diffeq_node(lab, tanh, input_var="li_op/I_ext", output_var="li_op/v", weights=bench.gen_rows(h, 0, h, dev, g=0.8),
                               source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32)
        for s_, t_ in zip(labs[:-1], labs[1:]):
            fb.add_edge(s_, t_, weights=bench.gen_rows(h, 0, h, dev, g=0.5))
        fb.add_edge(labs[-1], labs[0], weights=bench.gen_rows(h, 0, h, dev, g=0.5), feedback=True)
        x = torch.randn(200, h, device=dev) * 0.1
        fb.run(x, sampling_steps=10, verbose=False, enable_grad=False)
        big = torch.randn(8192, 8192, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            big = (big @ big).clamp_(-1, 1)
        t_queue = time.perf_counter() - t0
        t1 = time.perf_counter()
        fb.run(x, sampling_steps=10, verbose=False, enable_grad=False)
        t_call = time.perf_counter() - t1
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        out[f"{count} members"] = dict(queueing_the_long_work_ms=round(t_queue * 1e3, 2), host_time_of_run_ms=round(t_call * 1e3, 2),
                                       until_everything_finished_ms=round(t_all * 1e3, 2))
print(json.dumps(out))
