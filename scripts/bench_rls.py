"""RLS sweep at large n_in: achieved HBM bandwidth of the fused rank-1 update + next mat-vec pass (8 n^2 B/update)."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import sys

import torch

from rectipy_amd.rls import RLS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T, k = 20, 3
dev = torch.device("cuda:0")
edge = RLS(n, k, beta=0.99, alpha=1.0).to(dev)
edge.tiled = None if os.environ.get("RLS_TILED", "auto") == "auto" else os.environ["RLS_TILED"] == "1"
X = torch.randn(T, n, device=dev) * 0.1
Y = torch.randn(T, k, device=dev)
from rectipy_amd import _lib
lib = _lib.load()
if len(sys.argv) > 2:
    for R, U, nts in ((2, 4, 5), (1, 16, 5), (2, 8, 5), (2, 4, 1), (2, 4, 0), (2, 4, 3), (2, 4, 2), (1, 8, 1), (1, 16, 1), (1, 16, 0), (4, 4, 1), (2, 8, 1), (2, 8, 3),
                      (2, 4 + (512 << 8), 1), (2, 4 + (1024 << 8), 1), (2, 4 + (4096 << 8), 1), (2, 8 + (512 << 8), 1),
                      (1, 16 + (512 << 8), 1), (1, 16 + (256 << 8), 1)):
        lib.rp_set_rls_tuning(R, U, nts)
        edge.sweep(X[:3], Y[:3], 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        edge.sweep(X, Y, 1)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (T + 1)
        print(json.dumps(dict(R=R, U=U, nts=nts, ms=ms, GBps=8.0 * n * n / (ms * 1e-3) / 1e9)), flush=True)
    lib.rp_set_rls_tuning(0, 0, -1)
edge.sweep(X[:3], Y[:3], 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
edge.sweep(X, Y, 1)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / (T + 1)  # T fused passes + the initial mat-vec
print(json.dumps({"workload": f"RLS sweep n_in={n}, n_out={k}, {T} updates", "ms_per_update": ms,
                  "achieved_GBps": 8.0 * n * n / (ms * 1e-3) / 1e9, "frac_of_8TBps": 8.0 * n * n / (ms * 1e-3) / 1e9 / 8000,
                  "finite": bool(torch.isfinite(edge.P).all().item())}))
