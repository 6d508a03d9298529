"""Tile width and residency cap of the tiled RLS pass at n_in = 32768 (GPU box; benchmark-only knobs)."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rectipy_amd import _lib
from rectipy_amd.rls import RLS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T, k = 20, 3
dev = torch.device("cuda:0")
lib = _lib.load()
edge = RLS(n, k, beta=0.99, alpha=1.0).to(dev)
edge.tiled = True
X = torch.randn(T, n, device=dev) * 0.1
Y = torch.randn(T, k, device=dev)
res = []
for tu in (4, 2, 1):
    for per_cu in (0, 2, 3, 4, 5, 6, 8, 10, 12, 16):
        _lib.check(lib.rp_set_rls_tile(tu, per_cu))
        edge.sweep(X[:3], Y[:3], 1)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); edge.sweep(X, Y, 1); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (T + 1))
        res.append(dict(tile_kcols=tu, per_cu=per_cu, ms=round(best, 4), TBps=round(8.0 * n * n / best / 1e9, 3)))
        print(json.dumps(res[-1]), flush=True)
lib.rp_set_rls_tile(0, 0)
print("BEST", json.dumps(min(res, key=lambda r: r["ms"])))
