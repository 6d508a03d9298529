"""Where the tiled RLS pass spends its time at n_in = 32768: whole pass, pass without the mat-vec half (no x loads, no
reduction), wider tiles.  ms per update, TB/s of the 8 n^2 algorithmic bytes."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectipy_amd import _lib
from rectipy_amd.rls import RLS
lib = _lib.load()
n, T, k = 32768, 20, 3
dev = torch.device("cuda:0")
edge = RLS(n, k, beta=0.99, alpha=1.0).to(dev)
X = torch.randn(T, n, device=dev) * 0.1
Y = torch.randn(T, k, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for label, (R, U, nts), (tile, per_cu) in (("default", (0, 0, -1), (0, 0)), ("no mat-vec half", (0, 0, 5), (0, 0)),
                                           ("cached loads", (0, 0, 3), (0, 0)), ("plain stores", (0, 0, 0), (0, 0)),
                                           ("default again", (0, 0, -1), (0, 0))):
    lib.rp_set_rls_tuning(R, U, nts)
    lib.rp_set_rls_tile(tile, per_cu)
    edge.sweep(X[:3], Y[:3], 1)
    torch.cuda.synchronize()
    e0.record(); edge.sweep(X, Y, 1); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (T + 1)
    print(json.dumps({"variant": label, "ms_per_update": round(ms, 4), "TBps": round(8.0 * n * n / ms / 1e9, 3)}), flush=True)
lib.rp_set_rls_tuning(0, 0, -1)
lib.rp_set_rls_tile(0, 0)

for sym in (False, True, None):
    edge2 = RLS(n, k, beta=0.99, alpha=1.0).to(dev)
    edge2.symmetric = sym
    edge2.sweep(X[:5], Y[:5], 1)
    torch.cuda.synchronize()
    e0.record(); edge2.sweep(X, Y, 1); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (T + 1)
    print(json.dumps({"variant": f"symmetric={sym}", "ms_per_update_incl_mirror": round(ms, 4),
                      "TBps_of_8n2": round(8.0 * n * n / ms / 1e9, 3), "finite": bool(torch.isfinite(edge2.P).all().item())}), flush=True)
    del edge2
    torch.cuda.empty_cache()
