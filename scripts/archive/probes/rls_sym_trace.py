"""One symmetric RLS sweep per (rows per tile, rows in flight) under rocprofv3 --kernel-trace: per-kernel durations."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectipy_amd.rls import RLS
from rectipy_amd import _lib
lib = _lib.load()
n, T, k = 32768, 10, 3
dev = torch.device("cuda:0")
X = torch.randn(T, n, device=dev) * 0.1
Y = torch.randn(T, k, device=dev)
for rows, fl in ((32, 0),):
    edge = RLS(n, k, beta=0.99, alpha=1.0).to(dev)
    edge.symmetric = True
    edge.SYMMETRIC_ROWS = rows
    lib.rp_set_rls_sym(fl)
    edge.sweep(X, Y, 1)
    torch.cuda.synchronize()
    del edge
