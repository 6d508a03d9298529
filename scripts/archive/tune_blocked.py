"""Blocked vs per-row packed form of the sparse step kernel (tanh, GPU box): stored bytes and us per step; sweep of waves
per workgroup / groups in flight of the blocked kernel."""
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
prop = torch.cuda.get_device_properties(0)
print(json.dumps({"device": prop.name, "lds_per_block": getattr(prop, "shared_memory_per_block", None),
                  "lds_per_block_optin": getattr(prop, "shared_memory_per_block_optin", None)}), flush=True)


def timed(pop, x, T):
    pop.forward(50, x=x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pop.forward(T, x=x)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / T * 1e6, 1)


for N, p, T in ((8192, 0.1, 2000), (16384, 0.1, 1000), (32768, 0.1, 600)) if len(sys.argv) < 2 else ((int(sys.argv[1]), 0.1, 600),):
    gen = torch.Generator(device=dev).manual_seed(0)
    W = torch.randn(N, N, device=dev, generator=gen) * (1.5 / (p * N) ** 0.5)
    W *= torch.rand(N, N, device=dev, generator=gen) < p
    ideal = 6 * int((W != 0).sum().item())
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
    del W
    x = torch.zeros(T, 1, device=dev)
    res = {"N": N, "p": p, "ideal_MB": round(ideal / 1e6, 1)}
    pop.blocked_sparse = False
    res["per_row_MB"] = round(pop.pack_weights().stored_bytes / 1e6, 1)
    res["per_row_us"] = timed(pop, x, T)
    v_ref = pop.v.clone()
    pop.blocked_sparse = True
    for rb in ((4, 8, 16) if len(sys.argv) < 3 else (int(sys.argv[2]),)):
        pop.sparse_block_rows = rb
        res[f"rb{rb}_MB"] = round(pop.pack_weights().stored_bytes / 1e6, 1)
        for waves, U in ((16, 4), (16, 8), (8, 8), (16, 3), (16, 2), (16, 1), (8, 4), (8, 2)):
            lib.rp_set_blocked_tuning(waves, U)
            try:
                res[f"rb{rb}_w{waves}U{U}_us"] = timed(pop, x, T)
            except (RuntimeError, ValueError) as e:
                res[f"rb{rb}_w{waves}U{U}_us"] = "n/a"
    lib.rp_set_blocked_tuning(0, 0)
    print(json.dumps(res), flush=True)
    del pop
