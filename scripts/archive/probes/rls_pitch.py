"""Does the power-of-two row pitch of P (n_in = 32768 floats = 128 KiB) cost bandwidth?  ms per update of the symmetric
and the dense tiled RLS pass with P stored at a padded leading dimension (GPU box)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectipy_amd.rls import RLS
n, T, k = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 20, 3
dev = torch.device("cuda:0")
X = torch.randn(T, n, device=dev) * 0.1
Y = torch.randn(T, k, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for pad in (0, 64, 256, 1024, 1088, 4160):
    for sym in (True, False):
        edge = RLS(n, k, beta=0.99, alpha=1.0).to(dev)
        buf = torch.zeros(n, n + pad, device=dev)
        buf[:, :n].copy_(edge.P)
        edge.P = buf[:, :n]
        edge.symmetric = sym
        edge.sweep(X[:5], Y[:5], 1)
        torch.cuda.synchronize()
        e0.record(); edge.sweep(X, Y, 1); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (T + 1)
        print(json.dumps({"pad_floats": pad, "symmetric": sym, "ms_per_update": round(ms, 4),
                          "finite": bool(torch.isfinite(edge.P).all().item())}), flush=True)
        del edge, buf
        torch.cuda.empty_cache()
