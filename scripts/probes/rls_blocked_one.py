"""One blocked RLS sweep configuration at n_in = 32768 for rocprofv3: rows per tile, block, updates."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rectipy_amd.rls import RLS  # noqa: E402

dev = torch.device("cuda:0")
rows, block, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 64
n_in, k = 32768, 3
gen = torch.Generator(device=dev).manual_seed(5)
X = torch.randn(T, n_in, device=dev, generator=gen) * 0.1
Y = torch.randn(T, k, device=dev, generator=gen)
edge = RLS(n_in, k, beta=0.99, alpha=1.0).to(dev)
edge.symmetric, edge.block = True, block
edge.SYMMETRIC_ROWS, edge.BLOCK_ROWS = min(rows, 64), max(rows, 64)
edge.sweep(X[:16], Y[:16], 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
edge.sweep(X, Y, 1)
e1.record()
torch.cuda.synchronize()
print(round(e0.elapsed_time(e1) / T, 4), "ms per update")
