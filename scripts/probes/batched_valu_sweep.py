"""The vector-ALU batched step (bit-identical to unbatched runs) at N = 32768 on the measurement build: rows per wave x chunks in
flight per batch size (rp_set_batched_tuning), after the packed-fma / transposed-butterfly change."""
import json
import os
import sys

os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from rectipy_amd import _lib  # noqa: E402
from rectipy_amd.engine import BatchedPopulation, Population  # noqa: E402

dev = torch.device("cuda:0")
N, T = 32768, 200
W = bench.gen_rows(N, 0, N, dev)
pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
lib = _lib.load()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {}
combos = {2: [(0, 0), (2, 1), (2, 2), (4, 1), (4, 2)], 4: [(0, 0), (2, 1), (2, 2), (4, 1), (8, 1), (8, 2)],
          8: [(0, 0), (1, 1), (1, 2), (2, 1), (2, 2), (2, 4), (4, 1), (4, 2)], 16: [(0, 0), (1, 2), (1, 4), (2, 1)]}
for B, rus in combos.items():
    for R, U in rus:
        lib.rp_set_batched_tuning(R, U)
        bp = BatchedPopulation(pop, B, form="valu")
        bp.v.normal_()
        bp.refresh()
        x = torch.randn(T, B, 1, device=dev)
        try:
            bp.forward(20, x=x[:20])
            torch.cuda.synchronize()
            e0.record()
            bp.forward(T, x=x)
            e1.record()
            torch.cuda.synchronize()
            out[f"B{B}_R{R}_U{U}"] = round(e0.elapsed_time(e1) / T, 4)
        except Exception as exc:  # noqa: BLE001
            out[f"B{B}_R{R}_U{U}"] = str(exc)[:40]
        print(out, file=sys.stderr, flush=True)
lib.rp_set_batched_tuning(0, 0)
print(json.dumps(out))
