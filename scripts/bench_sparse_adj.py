"""us per step of the sparse forward step and of the adjoint on packed W^T, per-row vs blocked form (GPU box)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd import _abi  # noqa: E402
from rectipy_amd.engine import Population  # noqa: E402
dev = torch.device("cuda:0")
for N, p, T in ((16384, 0.1, 400), (32768, 0.1, 300)):
    gen = torch.Generator(device=dev).manual_seed(0)
    W = torch.randn(N, N, device=dev, generator=gen) * (1.5 / (p * N) ** 0.5)
    W *= torch.rand(N, N, device=dev, generator=gen) < p
    res = {"N": N, "p": p}
    for form in ("per_row", "blocked"):
        pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
        pop.blocked_sparse = form == "blocked"
        pk = pop.pack_weights()
        tv, tr, ut = (torch.empty(T, N, device=dev) for _ in range(3))
        g = torch.randn(T, N, device=dev, generator=gen) * 0.01
        x = torch.zeros(T, 1, device=dev)
        pop.forward(T, x=x, save_v=tv, save_src=tr)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pop.forward(T, x=x, save_v=tv, save_src=tr)
        torch.cuda.synchronize(); res[form + "_fwd_us"] = round((time.perf_counter() - t0) / T * 1e6, 1)
        lam = torch.zeros(N, device=dev)
        keep = pop.backward(tv, tr, g, _abi.RP_OUT_STATE0, lam, ut)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        keep = pop.backward(tv, tr, g, _abi.RP_OUT_STATE0, lam, ut)
        torch.cuda.synchronize(); res[form + "_adj_us"] = round((time.perf_counter() - t0) / T * 1e6, 1)
        res[form + "_block_rows"] = (pk.block_rows, pop.packed_t.block_rows)
        res[form + "_MB"] = round(pk.stored_bytes / 1e6, 1)
        del pop, keep
    print(json.dumps(res), flush=True)
