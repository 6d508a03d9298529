"""Forward / adjoint step time across network sizes (GPU box): where the path is launch-bound, cache-resident, HBM-bound."""
import json

import torch

from rectipy_amd import _abi
from rectipy_amd.engine import Population

dev = torch.device("cuda:0")
for N in (512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
    W = torch.randn(N, N, device=dev) * (1.5 / N ** 0.5)
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
    del W
    T = 2000 if N <= 4096 else (400 if N <= 16384 else 100)
    x = torch.zeros(T, 1, device=dev)
    Tb = min(T, 200 if N <= 16384 else 40)
    tv, tr, ut = (torch.zeros(Tb, N, device=dev) for _ in range(3))
    g = torch.randn(Tb, N, device=dev) * 0.01
    pop.forward(20, x=x)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record()
    pop.forward(T, x=x)
    e[1].record()
    pop.forward(Tb, x=x, save_v=tv, save_src=tr)
    lam = torch.zeros(N, device=dev)
    e[2].record()
    keep = pop.backward(tv, tr, g, _abi.RP_OUT_STATE0, lam, ut)
    e[3].record()
    torch.cuda.synchronize()
    fwd, bwd = e[0].elapsed_time(e[1]) / T, e[2].elapsed_time(e[3]) / Tb
    from rectipy_amd import _lib
    alt = {}
    for U, SW in ((8, 1), (16, 1), (6, 2), (4, 2)):
        _lib.load().rp_set_bwd_tuning(U, SW)
        lam = torch.zeros(N, device=dev)
        keep = pop.backward(tv, tr, g, _abi.RP_OUT_STATE0, lam, ut)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lam = torch.zeros(N, device=dev)
        a.record()
        keep = pop.backward(tv, tr, g, _abi.RP_OUT_STATE0, lam, ut)
        b.record()
        torch.cuda.synchronize()
        best = a.elapsed_time(b)
        for _ in range(2):
            lam = torch.zeros(N, device=dev)
            a.record()
            keep = pop.backward(tv, tr, g, _abi.RP_OUT_STATE0, lam, ut)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        alt[f"U{U}SW{SW}"] = round(best / Tb * 1e3, 1)
    _lib.load().rp_set_bwd_tuning(0, 0)
    print(json.dumps(dict(N=N, bwd_alt_us=alt, W_MB=4.0 * N * N / 1e6, fwd_us=fwd * 1e3, fwd_steps_per_s=1e3 / fwd,
                          fwd_GBps=4.0 * N * N / fwd / 1e6, bwd_us=bwd * 1e3, bwd_GBps=4.0 * N * N / bwd / 1e6)), flush=True)
    del pop, tv, tr, ut, g, keep
    torch.cuda.empty_cache()
