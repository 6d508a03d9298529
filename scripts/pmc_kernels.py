"""Workload for the per-kernel HBM-traffic passes (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE by
scripts/archive/profile_kernels.sh): a few launches each of the tanh step kernel, the adjoint strip kernel, the QIF step kernel
and the RLS pass at HBM-bound sizes.  Prints the algorithmic bytes per launch of each for the summary."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectipy_amd import _abi  # noqa: E402
from rectipy_amd.engine import Population  # noqa: E402
from rectipy_amd.rls import RLS  # noqa: E402

dev = torch.device("cuda:0")
N, T = 32768, 12
algo = {}

W = torch.randn(N, N, device=dev) * (1.5 / N ** 0.5)
pop = Population("li_tanh", W, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
tv, tr, ut = (torch.zeros(T, N, device=dev) for _ in range(3))
x = torch.zeros(T, 1, device=dev)
pop.forward(T, x=x, save_v=tv, save_src=tr)
g = torch.randn(T, N, device=dev) * 0.01
lam = torch.zeros(N, device=dev)
keep = pop.backward(tv, tr, g, _abi.RP_OUT_STATE0, lam, ut)
torch.cuda.synchronize()
algo["step_kernel<0"] = 4.0 * N * N + 4.0 * (9 * N + 1)          # W + v r/w, tau, k, eta, r r/w, two trajectory rows, input
algo["wt_partial_kernel"] = 4.0 * N * N + 4.0 * (N + 16 * N)     # W + u + the Q=16 chunk partials written
del pop, W, keep

Wq = torch.rand(N, N, device=dev) * (20.0 / N)
qif = Population("qif", Wq, dict(tau=1.0, k=1.0, eta=-5.0, tau_s=1.0), 1e-3, dev, v0=-2.0, s0=0.0)
qif.forward(T, x=torch.full((T, 1), 15.0, device=dev))
torch.cuda.synchronize()
algo["step_kernel<2"] = 4.0 * N * N + 4.0 * (8 * N + 1)
del qif, Wq

gen = torch.Generator(device=dev).manual_seed(0)
Ws = torch.randn(N, N, device=dev, generator=gen) * (1.5 / (0.1 * N) ** 0.5)
Ws *= torch.rand(N, N, device=dev, generator=gen) < 0.1
sp = Population("li_tanh", Ws, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
del Ws
packed = sp.pack_weights()
sp.forward(T, x=x)
torch.cuda.synchronize()
# packed stream + the source vector staged into LDS once per CU (256 x 128 KiB, served by L2) is not HBM traffic
algo["step_sparse_blocked_kernel"] = float(packed.stored_bytes) + 4.0 * 8 * N
del sp, packed

n_in = 16384
edge = RLS(n_in, 3, beta=0.99).to(dev)
edge.symmetric = False
X = torch.randn(6, n_in, device=dev) * 0.1
Y = torch.randn(6, 3, device=dev)
edge.sweep(X, Y, 1)
torch.cuda.synchronize()
algo["rls_tile_kernel"] = 8.0 * n_in * n_in + 8.0 * n_in * (n_in // 1024)                      # P read + written once per update
# r03: the symmetric pass streams the 1024 x 1024 blocks on and above the diagonal (read + written once per update) and
# writes one column partial per 64-row tile above the diagonal
edge2 = RLS(n_in, 3, beta=0.99).to(dev)
edge2.symmetric = True
edge2.sweep(X, Y, 1)
torch.cuda.synchronize()
nb = n_in // 1024
algo["rls_sym_kernel<true"] = 8.0 * 1024 * 1024 * nb * (nb + 1) / 2 + 4.0 * (n_in // 64) * n_in / 2
del edge, edge2
# r04: the blocked sweep's pass -- the same blocks read + written once per EIGHT updates, plus eight inputs' column partials
# (one per 128-row tile above the diagonal) and row partials
edge3 = RLS(n_in, 3, beta=0.99).to(dev)
edge3.symmetric, edge3.block = True, 8
X3 = torch.randn(32, n_in, device=dev) * 0.1
edge3.sweep(X3, torch.randn(32, 3, device=dev), 1)
torch.cuda.synchronize()
algo["rls_symk_kernel<8, true, true"] = (8.0 * 1024 * 1024 * nb * (nb + 1) / 2 + 8 * 4.0 * (n_in // edge3.BLOCK_ROWS) * n_in / 2
                                         + 8 * 4.0 * n_in * nb)
del edge3, X3

# r03: the batched matrix-core steps (W once per step for all trajectories; the source tiles come from L2)
import bench  # noqa: E402
from rectipy_amd.engine import BatchedPopulation  # noqa: E402
Wb = bench.gen_rows(N, 0, N, dev)
popb = Population("li_tanh", Wb, dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
for B in (16, 32):
    bp = BatchedPopulation(popb, B, mfma=True)
    bp.v.normal_()
    bp.refresh()
    bp.forward(T, x=torch.zeros(T, B, 1, device=dev))
    torch.cuda.synchronize()
    del bp
# W + per trajectory: v read/written, source read once per workgroup from L2 (not HBM) and written, parameters
algo["step_batched_mfma16_kernel<0, 1"] = 4.0 * N * N + 4.0 * 16 * (3 * N) + 12.0 * N
algo["step_batched_mfma16_split_kernel"] = 4.0 * N * N + 4.0 * 32 * (3 * N) + 12.0 * N
del popb, Wb
# r04: the coupled step (two populations of N/2 neurons, a -> b and feedback b -> a): per launch the target's rows of W and
# of its edge matrix, streamed by the same waves
from rectipy_amd.engine import forward_coupled  # noqa: E402
h = N // 2
arena = torch.randn(4 * h * h, device=dev) / h ** 0.5
mats = [arena[q * h * h:(q + 1) * h * h].view(h, h) for q in range(4)]
cp = [Population("li_tanh", mats[q], dict(tau=2.0, k=1.0, eta=0.0), 1e-2, dev) for q in range(2)]
hist = [torch.zeros(h, device=dev) for _ in range(3)]
forward_coupled([dict(pop=cp[0], run=dict(steps=T, x=torch.zeros(T, 1, device=dev), hist=hist), edges=[(mats[3], 1, _abi.RP_EDGE_LIVE)]),
                 dict(pop=cp[1], run=dict(steps=T), edges=[(mats[2], 0, _abi.RP_EDGE_OUT)])])
torch.cuda.synchronize()
algo["step_edges_kernel<0"] = 4.0 * h * (2 * h) + 4.0 * (9 * h + 2 * h)  # W rows + edge rows + vectors (two source vectors read)
del cp, mats, arena
# r05: the recursive filter of an edge with memory between two populations (rp_edge_mem: y <- F y + x, F as large as a
# recurrent matrix; the recurrent row-dot kernel with the filter's epilogue), and the adjoint strips at the coupled fit's
# matrix size (16384 x 16384: 2-KiB strips x 2 rows in flight since r05)
from rectipy_amd.engine import EdgeMemory  # noqa: E402
Ff = bench.gen_rows(h, 0, h, dev, g=0.5)
mem = EdgeMemory("filter", h, dev, F=Ff)
mem.window(torch.randn(T, h, device=dev) * 0.1)
torch.cuda.synchronize()
algo["step_kernel<65"] = 4.0 * h * h + 4.0 * 3 * h       # F once + y read, x read, y' written
del mem, Ff
pa = Population("li_tanh", bench.gen_rows(h, 0, h, dev), dict(tau=10.0, k=1.0, eta=0.0), 1e-2, dev)
tv2, tr2, ut2 = (torch.zeros(T, h, device=dev) for _ in range(3))
pa.forward(T, x=torch.zeros(T, 1, device=dev), save_v=tv2, save_src=tr2)
keep2 = pa.backward(tv2, tr2, torch.randn(T, h, device=dev) * 0.01, _abi.RP_OUT_STATE0, torch.zeros(h, device=dev), ut2)
torch.cuda.synchronize()
algo["wt_partial_kernel<true, 2, 2>"] = 4.0 * h * h + 4.0 * (h + 48 * h)  # the matrix + u + the Q = 48 chunk partials written
del pa, keep2
print(json.dumps({"algorithmic_bytes_per_launch": algo}), flush=True)
