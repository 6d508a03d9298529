"""Walks of the dense step kernel at mid sizes (measurement build: rp_set_tuning): rows per wave x chunks in flight x workgroups."""
import json
import os
import sys

os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from rectipy_amd import _lib  # noqa: E402
from rectipy_amd.engine import Population  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {}
for N in (4096, 6144, 8192):
    W = bench.gen_rows(N, 0, N, dev)
    T = 1500
    x = torch.zeros(T, 1, device=dev)
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev, tune=False)
    best = None
    for R, U in ((0, 0), (1, 4), (1, 8), (2, 1), (2, 2), (2, 4), (4, 1), (4, 2), (8, 1)):
        for blocks in ((0,) if R == 0 else (0, 256, 512, 768, 1024, 2048)):
            lib.rp_set_tuning(R, U, -1, blocks)
            try:
                pop.forward(40, x=x[:40])
                torch.cuda.synchronize()
                e0.record()
                pop.forward(T, x=x)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / T * 1e3
            except Exception as exc:  # noqa: BLE001
                us = float("nan")
            out[f"N{N}_R{R}_U{U}_b{blocks}"] = round(us, 2)
            if R and (best is None or us < best[0]):
                best = (us, R, U, blocks)
    out[f"N{N}_best"] = best
    print({k: v for k, v in out.items() if k.startswith(f"N{N}")}, file=sys.stderr, flush=True)
    del W
    torch.cuda.empty_cache()
lib.rp_set_tuning(0, 0, -1, 0)
print(json.dumps({k: v for k, v in out.items() if k.endswith("best") or "_R0_" in k}))
