"""Streaming configuration (rows per wave R, chunks in flight U, resident workgroups) of the tanh step kernel over the
shapes the benches use; candidates are the grids that divide the row groups into a whole number of passes (GPU box only).
usage: sweep_forward_shapes.py "ROWSxN;ROWSxN;..." [steps] [grid,grid,...]"""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json

import torch

from rectipy_amd import _lib
from rectipy_amd.engine import Population

shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1].split(";")]
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 40
GRIDS = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
lib = _lib.load()
os.environ["RECTIPY_AMD_NO_AUTOTUNE"] = "1"  # the sweep itself runs on the static rule
if os.environ.get('CFGS'):
    valid = [tuple(int(v) for v in c.split(',')) for c in os.environ['CFGS'].split(';')]
else:
    valid = [(1, 1), (1, 2), (1, 4), (1, 8), (3, 1), (2, 1), (2, 2), (2, 4), (4, 1), (4, 2), (4, 4), (8, 1), (8, 2)]
for rows, n in shapes:
    W = torch.empty(rows, n, device=dev).normal_(0.0, 1.5 / n ** 0.5)
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev, n=n, row0=0)
    del W
    pop.source.copy_(torch.tanh(torch.randn(n, device=dev)))
    x = torch.zeros(STEPS, 1, device=dev)
    cands = [(0, 0, 0)]
    for R, U in valid:
        groups = (rows + R - 1) // R
        for mb in GRIDS or [-(-groups // (4 * passes)) for passes in (1, 2, 3, 4, 6, 8)]:
            if mb >= 128 and mb * 4 <= groups and (R, U, mb) not in cands:
                cands.append((R, U, mb))
    res = []
    for R, U, mb in cands:
        lib.rp_set_tuning(R, U, 1, mb)
        pop.forward(3, x=x)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pop.forward(STEPS, x=x)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / STEPS)
        res.append(dict(rows=rows, n=n, R=R, U=U, max_blocks=mb, us=round(best * 1e3, 2), TBps=round(4.0 * rows * n / best / 1e9, 3)))
    lib.rp_set_tuning(0, 0, 1, 0)
    res.sort(key=lambda r: r["us"])
    print(json.dumps(dict(shape=[rows, n], built_in=[r for r in res if r["R"] == 0][0])), flush=True)
    for r in res[:int(os.environ.get('TOP', 8))]:
        print(json.dumps(r), flush=True)
    # what the in-library measurement (rp_tune_forward) picks for this shape, and what a run then costs
    import ctypes as C
    import time
    t0 = time.perf_counter()
    os.environ["RECTIPY_AMD_NO_AUTOTUNE"] = "0"
    pop.backend.tune_forward(pop._net(), dev)
    os.environ["RECTIPY_AMD_NO_AUTOTUNE"] = "1"
    torch.cuda.synchronize()
    tune_s = time.perf_counter() - t0
    pop.forward(3, x=x)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        pop.forward(STEPS, x=x)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / STEPS)
    print(json.dumps(dict(shape=[rows, n], tuned=pop.backend.forward_tuning(pop._net()), tune_seconds=round(tune_s, 3),
                          us=round(best * 1e3, 2), TBps=round(4.0 * rows * n / best / 1e9, 3))), flush=True)
    del pop
    torch.cuda.empty_cache()
