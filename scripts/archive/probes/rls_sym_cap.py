"""Symmetric RLS pass: (rows per tile, rows in flight) x residency cap (workgroups per CU, through an LDS allocation);
ms per update incl. the small kernels and the mirror's share (GPU box)."""
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
cases = [(32, 4, 0), (32, 4, 1), (32, 4, 15), (32, 8, 15), (64, 4, 1), (16, 4, 1), (32, 4, 2), (32, 2, 1), (32, 8, 1)]  # (0: built-in cap, 15: none)
edges = {}
for rows in (16, 32, 64):
    edges[rows] = RLS(n, k, beta=0.99, alpha=1.0).to(dev)
    edges[rows].symmetric = True
    edges[rows].SYMMETRIC_ROWS = rows
    if rows != 64:
        edges[rows].P = edges[64].P if 64 in edges else edges[rows].P
P = torch.eye(n, device=dev)
for e in edges.values():
    e.P = P  # one matrix for all three (4 GiB each otherwise)
for rep in range(3):
    for rows, fl, per_cu in cases:
        edge = edges[rows]
        lib.rp_set_rls_sym(fl | (per_cu << 8))
        P.zero_(); P.diagonal().fill_(1.0)
        edge.sweep(X[:3], Y[:3], 1)
        torch.cuda.synchronize()
        e0.record(); edge.sweep(X, Y, 1); e1.record(); torch.cuda.synchronize()
        print(json.dumps({"rep": rep, "rows_per_tile": rows, "rows_in_flight": fl, "workgroups_per_cu_cap": per_cu,
                          "ms_per_update": round(e0.elapsed_time(e1) / (T + 1), 4)}), flush=True)
lib.rp_set_rls_sym(0)
