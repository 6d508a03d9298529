"""Coupled step: matrices from four separate allocations (as a front end gets them) against one arena, with and without
allocator churn before them.  Engine level, events, 200-step calls."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from rectipy_amd import _abi  # noqa: E402
from rectipy_amd.engine import Population, forward_coupled  # noqa: E402

dev = torch.device("cuda:0")
T, N = 200, 32768
h = N // 2
gen = torch.Generator(device=dev).manual_seed(1)


def timed(mats, label):
    pops = [Population("li_tanh", mats[q], dict(tau=2.0, k=1.0, eta=0.0), 1e-2, dev) for q in range(2)]
    xh = torch.randn(T, h, device=dev, generator=gen) * 0.1
    hist = [torch.zeros(h, device=dev) for _ in range(3)]

    def run(steps):
        forward_coupled([dict(pop=pops[0], run=dict(steps=steps, x=xh[:steps], hist=hist), edges=[(mats[3], 1, _abi.RP_EDGE_LIVE)]),
                         dict(pop=pops[1], run=dict(steps=steps), edges=[(mats[2], 0, _abi.RP_EDGE_OUT)])])
    run(20)
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(T)
        e1.record()
        torch.cuda.synchronize()
        res.append(round(e0.elapsed_time(e1) / T * 1e3, 1))
    print(label, res, [hex(m.data_ptr()) for m in mats], flush=True)
    return res


out = {}
mats = [torch.randn(h, h, device=dev, generator=gen) / h ** 0.5 for _ in range(4)]
out["separate_fresh_process"] = timed(mats, "separate, fresh process")
del mats
torch.cuda.empty_cache()
mats = [bench.gen_rows(h, 0, h, dev, g=g) for g in (1.5, 1.5, 0.5, 0.5)]
out["gen_rows"] = timed(mats, "bench.gen_rows")
del mats
torch.cuda.empty_cache()
# churn: many mid-sized allocations freed again, then the matrices
junk = [torch.empty(int(3e6) + 4096 * i, device=dev) for i in range(400)]
del junk
torch.cuda.empty_cache()
mats = [torch.randn(h, h, device=dev, generator=gen) / h ** 0.5 for _ in range(4)]
out["separate_after_churn"] = timed(mats, "separate, after churn")
del mats
torch.cuda.empty_cache()
arena = torch.randn(4 * h * h, device=dev, generator=gen) / h ** 0.5
mats = [arena[q * h * h:(q + 1) * h * h].view(h, h) for q in range(4)]
out["arena"] = timed(mats, "arena")
print(json.dumps(out))
