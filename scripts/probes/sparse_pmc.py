"""Workload for the counter passes of the packed sparse step (scripts/profile/sparse_pmc.sh): N = 32768, p = 0.1, the blocked
form (8-row blocks) and, for comparison, the per-row form; 30 steps each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rectipy_amd.engine import Population  # noqa: E402

dev = torch.device("cuda:0")
N, p, T = 32768, 0.1, 30
gen = torch.Generator(device=dev).manual_seed(0)
W = torch.randn(N, N, device=dev, generator=gen) * (1.5 / (p * N) ** 0.5)
W *= torch.rand(N, N, device=dev, generator=gen) < p
x = torch.zeros(T, 1, device=dev)
for blocked in (True, False):
    pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
    pop.blocked_sparse = blocked
    packed = pop.pack_weights()
    pop.forward(T, x=x)
    torch.cuda.synchronize()
    print(("blocked" if blocked else "per-row"), "stored bytes", packed.stored_bytes, flush=True)
    del pop
