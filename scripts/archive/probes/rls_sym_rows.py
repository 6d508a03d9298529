"""ms per update of the RLS passes at n_in = 32768: dense tiled vs symmetric with 8 / 16 / 32 rows per tile (GPU box)."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectipy_amd.rls import RLS
n, T, k = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 20, 3
dev = torch.device("cuda:0")
X = torch.randn(T, n, device=dev) * 0.1
Y = torch.randn(T, k, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
from rectipy_amd import _lib
lib = _lib.load()
for sym, rows, fl in ((False, 0, 0), (True, 16, 2), (True, 16, 4), (True, 16, 8), (True, 32, 2), (True, 32, 4), (True, 32, 8),
                      (True, 64, 1), (True, 64, 2), (True, 64, 4), (True, 64, 8), (False, 0, 0)):
    edge = RLS(n, k, beta=0.99, alpha=1.0).to(dev)
    edge.symmetric = sym
    lib.rp_set_rls_sym(fl)
    if rows:
        edge.SYMMETRIC_ROWS = rows
    edge.sweep(X[:5], Y[:5], 1)
    torch.cuda.synchronize()
    e0.record(); edge.sweep(X, Y, 1); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (T + 1)
    print(json.dumps({"symmetric": sym, "rows_per_tile": rows, "rows_in_flight": fl, "ms_per_update_incl_mirror": round(ms, 4),
                      "TBps_of_8n2": round(8.0 * n * n / ms / 1e9, 3), "finite": bool(torch.isfinite(edge.P).all().item())}), flush=True)
    del edge
    torch.cuda.empty_cache()
lib.rp_set_rls_sym(0)
