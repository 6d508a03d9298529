"""Forward / adjoint step time across network sizes (GPU box): where the path is launch-bound, cache-resident, HBM-bound.
Adjoint: the default layout (fused row-dot launch on the transposed copy up to RP_BWD_TRANSPOSE_MAX_N, column strips
above; the per-sweep transposition is inside the timed call) and the column-strip kernels at every size."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd import _abi  # noqa: E402
from rectipy_amd.engine import Population  # noqa: E402

dev = torch.device("cuda:0")
for N in (512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
    W = torch.randn(N, N, device=dev) * (1.5 / N ** 0.5)
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
    del W
    T = 2000 if N <= 4096 else (400 if N <= 16384 else 100)
    x = torch.zeros(T, 1, device=dev)
    Tb = min(T, 1000 if N <= 4096 else (200 if N <= 16384 else 40))
    tv, tr, ut = (torch.zeros(Tb, N, device=dev) for _ in range(3))
    g = torch.randn(Tb, N, device=dev) * 0.01
    pop.forward(20, x=x)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record()
    pop.forward(T, x=x)
    e[1].record()
    torch.cuda.synchronize()
    fwd = e[0].elapsed_time(e[1]) / T
    pop.forward(Tb, x=x, save_v=tv, save_src=tr)

    def bwd_time(transposed):
        pop.transposed_adjoint = transposed
        best = None
        for _ in range(3):
            lam = torch.zeros(N, device=dev)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            keep = pop.backward(tv, tr, g, _abi.RP_OUT_STATE0, lam, ut)
            b.record()
            torch.cuda.synchronize()
            del keep
            t = a.elapsed_time(b) / Tb
            best = t if best is None else min(best, t)
        return best

    bwd = bwd_time(True)
    strip = bwd_time(False) if N <= _abi.RP_BWD_TRANSPOSE_MAX_N else bwd
    print(json.dumps(dict(N=N, W_MB=round(4.0 * N * N / 1e6, 1), fwd_us=round(fwd * 1e3, 2),
                          fwd_GBps=round(4.0 * N * N / fwd / 1e6, 1), bwd_us=round(bwd * 1e3, 2),
                          bwd_GBps=round(4.0 * N * N / bwd / 1e6, 1), bwd_strip_us=round(strip * 1e3, 2))), flush=True)
    del pop, tv, tr, ut, g
    torch.cuda.empty_cache()
