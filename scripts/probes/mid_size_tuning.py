"""Static walk against the measured one (rp_tune_forward) at mid sizes: us per step of rp_forward, tanh population."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from rectipy_amd.engine import Population  # noqa: E402

dev = torch.device("cuda:0")
out = {}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for N in (2048, 3072, 4096, 6144, 8192, 12288, 16384):
    W = bench.gen_rows(N, 0, N, dev)
    T = 2000 if N <= 8192 else 600
    x = torch.zeros(T, 1, device=dev)
    for tune in (False, True):
        pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev, tune=tune)
        pop.forward(50, x=x[:50])
        torch.cuda.synchronize()
        e0.record()
        pop.forward(T, x=x)
        e1.record()
        torch.cuda.synchronize()
        out[f"N{N}_{'tuned' if tune else 'static'}"] = round(e0.elapsed_time(e1) / T * 1e3, 2)
        if tune:
            out[f"N{N}_cfg"] = pop.backend.forward_tuning(pop._net())
        print(out, file=sys.stderr, flush=True)
    del W
    torch.cuda.empty_cache()
print(json.dumps(out, default=str))
