"""Sweep rows per wave / groups in flight of the sparse step kernel (tanh, GPU box)."""
import os
os.environ.setdefault("RECTIPY_AMD_MEASUREMENT", "1")  # the rp_set_* knobs live in the measurement build (python -m rectipy_amd.build --measurement)
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd import _lib  # noqa: E402
from rectipy_amd.engine import Population  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
for N, p, T in ((2048, 0.1, 3000), (4096, 0.1, 3000), (8192, 0.1, 2000), (16384, 0.1, 1000), (32768, 0.1, 600), (32768, 0.2, 400)):
    gen = torch.Generator(device=dev).manual_seed(0)
    W = torch.randn(N, N, device=dev, generator=gen) * (1.5 / (p * N) ** 0.5)
    W *= torch.rand(N, N, device=dev, generator=gen) < p
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
    del W
    pop.pack_weights()
    x = torch.zeros(T, 1, device=dev)
    res = {"N": N, "p": p}
    for R, U in ((1, 4), (1, 8), (2, 2), (2, 4), (2, 8), (4, 2)):
        lib.rp_set_sparse_tuning(R, U)
        pop.forward(50, x=x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pop.forward(T, x=x)
        torch.cuda.synchronize()
        res[f"R{R}U{U}_us"] = round((time.perf_counter() - t0) / T * 1e6, 1)
    lib.rp_set_sparse_tuning(0, 0)
    print(json.dumps(res), flush=True)
    del pop
