"""Where does the symmetric RLS pass lose its time?  The (32 rows, 8 in flight) kernel with parts removed and with the
residency capped through an LDS allocation; ms per update incl. the small kernels (GPU box; results of probe runs are wrong
by construction)."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectipy_amd.rls import RLS
from rectipy_amd import _lib
lib = _lib.load()
n, T, k = 32768, 20, 3
dev = torch.device("cuda:0")
X = torch.randn(T, n, device=dev) * 0.1
Y = torch.randn(T, k, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
edge = RLS(n, k, beta=0.99, alpha=1.0).to(dev)
edge.symmetric = True
edge.SYMMETRIC_ROWS = 32
for probe in (0, 1, 2, 3):
    for per_cu in (15, 1, 2, 3, 4, 6, 8):
        if per_cu == 15 and probe == 0:
            lib.rp_set_rls_sym(8 | (15 << 8))
        else:
            lib.rp_set_rls_sym(8 | (per_cu << 8) | (probe << 12))
        edge.P.copy_(torch.eye(n, device=dev))
        edge.sweep(X[:3], Y[:3], 1)
        torch.cuda.synchronize()
        e0.record(); edge.sweep(X, Y, 1); e1.record(); torch.cuda.synchronize()
        print(json.dumps({"probe": probe, "workgroups_per_cu_cap": per_cu, "ms_per_update": round(e0.elapsed_time(e1) / (T + 1), 4)}), flush=True)
lib.rp_set_rls_sym(0)
