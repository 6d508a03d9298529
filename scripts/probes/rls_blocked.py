"""RLS sweep at n_in = 32768: ms per update for one pass per update (block = 1) and the blocked sweep (2, 4, 8), by rows
per tile; 64 consecutive updates."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rectipy_amd.rls import RLS  # noqa: E402

dev = torch.device("cuda:0")
n_in, k, T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 3, 64
gen = torch.Generator(device=dev).manual_seed(5)
X = torch.randn(T, n_in, device=dev, generator=gen) * 0.1
Y = torch.randn(T, k, device=dev, generator=gen)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {}
for rows in (64, 128):
    for block in (1, 2, 4, 8):
        edge = RLS(n_in, k, beta=0.99, alpha=1.0).to(dev)
        edge.symmetric, edge.block = True, block
        edge.SYMMETRIC_ROWS, edge.BLOCK_ROWS = min(rows, 64), rows
        edge.sweep(X[:16], Y[:16], 1)
        torch.cuda.synchronize()
        e0.record()
        edge.sweep(X, Y, 1)
        e1.record()
        torch.cuda.synchronize()
        out[f"rows{rows}_block{block}"] = round(e0.elapsed_time(e1) / T, 4)
        print(out, file=sys.stderr, flush=True)
        del edge
        torch.cuda.empty_cache()
print(json.dumps(out))
