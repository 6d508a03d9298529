"""Walk of W for the vector-ALU batched step (rows per wave, chunks in flight, resident workgroups) at N = 32768."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rectipy_amd import _lib  # noqa: E402
from rectipy_amd.engine import BatchedPopulation, Population  # noqa: E402
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = 60
lib = _lib.load()
pop = Population("li_tanh", bench.gen_rows(N, 0, N, dev), dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cases = {2: [(0, 0), (2, 1), (2, 2), (4, 1), (4, 2)], 4: [(0, 0), (2, 1), (2, 2), (4, 1), (8, 1), (8, 2)],
         8: [(0, 0), (2, 1), (4, 1), (1, 1), (1, 2), (2, 2)]}
for B, cfgs in cases.items():
    bp = BatchedPopulation(pop, B, mfma=False)
    bp.v.normal_()
    bp.refresh()
    x = torch.randn(T, B, 1, device=dev)
    res = []
    for R, U in cfgs:
        for grid in (0, 512, 683, 768, 1024, 1366):
            if R == 0 and grid:
                continue
            lib.rp_set_batched_tuning(R, U)
            lib.rp_set_batched_grid(grid)
            bp.forward(5, x=x[:5])
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                e0.record(); bp.forward(T, x=x); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / T)
            res.append(dict(B=B, R=R, U=U, grid=grid, ms=round(best, 4)))
    lib.rp_set_batched_tuning(0, 0)
    lib.rp_set_batched_grid(0)
    res.sort(key=lambda r: r["ms"])
    print(json.dumps([r for r in res if r["R"] == 0][0]), flush=True)
    for r in res[:6]:
        print(json.dumps(r), flush=True)
