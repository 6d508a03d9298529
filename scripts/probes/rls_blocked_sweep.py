"""Blocked RLS pass at n_in = 32768 on the measurement build: rows in flight x residency cap x rows per tile x block."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rectipy_amd.engine import HipBackend  # noqa: E402
from rectipy_amd.rls import RLS  # noqa: E402

dev = torch.device("cuda:0")
mb = HipBackend(measurement=True)
n_in, k, T = 32768, 3, 64
gen = torch.Generator(device=dev).manual_seed(5)
X = torch.randn(T, n_in, device=dev, generator=gen) * 0.1
Y = torch.randn(T, k, device=dev, generator=gen)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {}
for block in (8,):
    for rows in (128, 256):
        edge = RLS(n_in, k, beta=0.99, alpha=1.0, backend=mb).to(dev)
        edge.symmetric, edge.block, edge.BLOCK_ROWS = True, block, rows
        for R in (2, 4):
            for cap in ((2, 0) if rows <= 128 else (0,)):
                mb.lib.rp_set_rls_blk(R, cap)
                edge.sweep(X[:16], Y[:16], 1)
                torch.cuda.synchronize()
                e0.record()
                edge.sweep(X, Y, 1)
                e1.record()
                torch.cuda.synchronize()
                out[f"K{block}_rows{rows}_R{R}_cap{cap}"] = round(e0.elapsed_time(e1) / T, 4)
                print(out, file=sys.stderr, flush=True)
        del edge
        torch.cuda.empty_cache()
mb.lib.rp_set_rls_blk(0, -1)
print(json.dumps(out))
