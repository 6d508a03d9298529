"""Packed sparse forward at n > 32768 (GPU box): windowed LDS staging vs the plain packed form with L2 gathers vs the
dense kernel, N = 65536 and a 16384 x 262144 row shard (the per-GPU column count of BASELINE config 5)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd.engine import Population  # noqa: E402

dev = torch.device("cuda:0")
for rows, N, p, T in ((65536, 65536, 0.1, 150), (65536, 65536, 0.05, 200), (16384, 262144, 0.1, 150)):
    gen = torch.Generator(device=dev).manual_seed(0)
    W = torch.empty(rows, N, device=dev)
    for r0 in range(0, rows, 2048):
        blk = torch.randn(2048, N, device=dev, generator=gen) * (1.5 / (p * N) ** 0.5)
        blk *= torch.rand(2048, N, device=dev, generator=gen) < p
        W[r0:r0 + 2048] = blk
    del blk
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev, n=N, row0=0)
    del W
    x = torch.zeros(T, 1, device=dev)

    def rate():
        pop.forward(10, x=x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pop.forward(T, x=x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / T * 1e6

    res = {"rows": rows, "N": N, "p": p, "dense_us": round(rate(), 1)}
    for label, windowed in (("windowed", True), ("l2_gather", False)):
        pop.windowed_sparse = windowed
        t0 = time.perf_counter()
        packed = pop.pack_weights()
        torch.cuda.synchronize()
        res[label + "_pack_s"] = round(time.perf_counter() - t0, 2)
        res[label + "_us"] = round(rate(), 1)
        res[label + "_stored_MB"] = round(packed.stored_bytes / 1e6, 1)
    res["dense_MB"] = round(4.0 * rows * N / 1e6, 1)
    print(json.dumps(res), flush=True)
    del pop, packed
    torch.cuda.empty_cache()
