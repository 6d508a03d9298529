"""Blocked packed step at N = 32768, p = 0.1: us per step for (waves per workgroup, groups per register set) of the
measurement build -- since r05 two register sets are in flight per wave (2 U groups)."""
import json
import os
import sys

os.environ["RECTIPY_AMD_MEASUREMENT"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rectipy_amd.engine import Population  # noqa: E402

dev = torch.device("cuda:0")
N, p, T = 32768, 0.1, 200
gen = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(N, N, device=dev, generator=gen) * (1.5 / (p * N) ** 0.5)
W *= torch.rand(N, N, device=dev, generator=gen) < p
x = torch.zeros(T, 1, device=dev)
res = {}
for rb in (8, 4):
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
    pop.sparse_block_rows = rb
    packed = pop.pack_weights()
    lib = pop.backend.lib
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for waves in (16, 8):
        for U in (0, 1, 2, 3, 4):
            lib.rp_set_blocked_tuning(waves if U else 0, U)
            try:
                pop.forward(20, x=x)
                e0.record()
                pop.forward(T, x=x)
                e1.record()
                torch.cuda.synchronize()
                res[f"rb{rb}_waves{waves}_U{U}"] = round(e0.elapsed_time(e1) / T * 1e3, 1)
            except Exception as e:
                res[f"rb{rb}_waves{waves}_U{U}"] = str(e)[:60]
    lib.rp_set_blocked_tuning(0, 0)
    res[f"rb{rb}_stored_MB"] = round(packed.stored_bytes / 1e6, 1)
    del pop
print(json.dumps(res, indent=1))
