"""Packed sparse coupling matrix vs dense streaming (GPU box): steps/s of the tanh step at N = 32768 (LDS gather) and
N = 65536 (L2 gather) for the connection probabilities RectiPy's examples use (utility.random_connectivity, p = 0.1)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd.engine import Population  # noqa: E402

dev = torch.device("cuda:0")
for N, T in ((8192, 2000), (32768, 600)):
    for p in (0.05, 0.1, 0.2):
        gen = torch.Generator(device=dev).manual_seed(0)
        W = torch.empty(N, N, device=dev)
        for r0 in range(0, N, 4096):  # built in blocks: the mask would double the footprint at N = 65536
            blk = torch.randn(4096, N, device=dev, generator=gen) * (1.5 / (p * N) ** 0.5)
            blk *= torch.rand(4096, N, device=dev, generator=gen) < p
            W[r0:r0 + 4096] = blk
        del blk
        pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
        del W
        x = torch.sin(torch.arange(T, device=dev) * 0.01).reshape(-1, 1).contiguous()

        def rate():
            pop.forward(20, x=x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pop.forward(T, x=x)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / T

        dense = rate()
        v_dense = pop.v.clone()
        pop.v.zero_(); pop.refresh()
        t0 = time.perf_counter()
        packed = pop.pack_weights()
        torch.cuda.synchronize()
        t_pack = time.perf_counter() - t0
        sparse = rate()
        pop.v.zero_(); pop.refresh(); pop.pack_weights(False)
        pop.forward(20 + T, x=torch.cat([x[:20], x]))
        pop.pack_weights(); pop.v.zero_(); pop.refresh()
        pop.forward(20 + T, x=torch.cat([x[:20], x]))
        # adjoint sweep: dense column strips vs packed W^T
        Tb = min(T, 100)
        tv, ts, ut = (torch.zeros(Tb, N, device=dev) for _ in range(3))
        pop.forward(Tb, x=x, save_v=tv, save_src=ts)
        g = torch.randn(Tb, N, device=dev) * 0.01

        def bwd(sparse_adj):
            pop.sparse_adjoint = sparse_adj
            best = 1e9
            for _ in range(3):
                lam = torch.zeros(N, device=dev)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                keep = pop.backward(tv, ts, g, 0, lam, ut)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t1) / Tb)
                del keep
            return best

        bwd_sparse, bwd_dense = bwd(True), bwd(False)
        del tv, ts, ut, g
        print(json.dumps(dict(N=N, p=p, dense_us=round(dense * 1e6, 1), sparse_us=round(sparse * 1e6, 1),
                              bwd_dense_us=round(bwd_dense * 1e6, 1), bwd_sparse_us=round(bwd_sparse * 1e6, 1),
                              speedup=round(dense / sparse, 2), stored_MB=round(packed.stored_bytes / 1e6, 1),
                              dense_MB=round(4.0 * N * N / 1e6, 1), stream_GBps=round(packed.stored_bytes / sparse / 1e9, 1),
                              pack_s=round(t_pack, 2))), flush=True)
        del pop, packed
        torch.cuda.empty_cache()
