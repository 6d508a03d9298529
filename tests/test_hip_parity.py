"""HIP kernels vs the CPU oracle through the C ABI -- bit-exact (same summation tree, same
roundings).  Sizes are small enough for the oracle to finish in seconds."""
import os

import numpy as np
import pytest
import torch

from rectipy_amd import _abi
from rectipy_amd.engine import Population

pytestmark = pytest.mark.gpu


def make_li(N, seed, model="li_tanh", g=1.5):
    gen = torch.Generator().manual_seed(seed)
    W = torch.randn(N, N, generator=gen) * g / N ** 0.5
    params = dict(tau=torch.rand(N, generator=gen) * 5 + 5, k=torch.rand(N, generator=gen) + 0.5,
                  eta=torch.randn(N, generator=gen) * 0.1)
    if model == "li_sigmoid":
        params.update(r_max=torch.rand(N, generator=gen) + 0.5, s=torch.rand(N, generator=gen) * 2 + 0.5,
                      v0=torch.randn(N, generator=gen) * 0.2)
    v0 = torch.randn(N, generator=gen)
    return W, params, v0


def both(oracle_backend, hip_device, model, W, params, dt, **kw):
    cpu = Population(model, W, params, dt, "cpu", backend=oracle_backend, **kw)
    gpu = Population(model, W, params, dt, hip_device, **kw)
    return cpu, gpu


def assert_same(a, b, what):
    a = a.detach().cpu().numpy()
    b = b.detach().cpu().numpy()
    assert a.shape == b.shape, what
    same = (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))
    assert same.all(), f"{what}: {np.count_nonzero(~same)} of {a.size} differ, max abs {np.abs(a - b).max()}"


@pytest.mark.parametrize("N", [10, 64, 250, 256, 1000, 1024, 2052])
@pytest.mark.parametrize("model", ["li_tanh", "li_sigmoid"])
def test_rate_forward_bit_exact(oracle_backend, hip_device, N, model):
    W, params, v0 = make_li(N, seed=N, model=model)
    T = 40
    x = torch.randn(T, N, generator=torch.Generator().manual_seed(1)) * 0.5
    cpu, gpu = both(oracle_backend, hip_device, model, W, params, 1e-2, v0=v0)
    assert_same(cpu.source, gpu.source, "initial source")
    outs = []
    for pop, dev in ((cpu, "cpu"), (gpu, hip_device)):
        traj = torch.zeros(T, N, device=dev)
        sv = torch.zeros(T, N, device=dev)
        sr = torch.zeros(T, N, device=dev)
        sa = torch.zeros(T, N, device=dev)
        pop.forward(T, x=x.to(dev), traj=traj, save_v=sv, save_src=sr, save_acc=sa, out_var=_abi.RP_OUT_SOURCE)
        outs.append((traj, sv, sr, sa, pop.v, pop.source))
    for a, b, name in zip(outs[0], outs[1], ("traj", "save_v", "save_src", "save_acc", "v", "source")):
        assert_same(a, b, f"{model} N={N} {name}")


@pytest.mark.parametrize("mode", ["none", "bcast", "proj"])
def test_input_modes_and_sampling(oracle_backend, hip_device, mode):
    N, T, m = 512, 57, 3
    W, params, v0 = make_li(N, seed=3)
    gen = torch.Generator().manual_seed(5)
    x = {"none": None, "bcast": torch.randn(T, 1, generator=gen), "proj": torch.randn(T, m, generator=gen)}[mode]
    w_in = torch.randn(N, m, generator=gen) if mode == "proj" else None
    cpu, gpu = both(oracle_backend, hip_device, "li_tanh", W, params, 1e-2, v0=v0)
    res = []
    for pop, dev in ((cpu, "cpu"), (gpu, hip_device)):
        S = 40
        samples = torch.zeros(S, N, device=dev)
        rec = torch.zeros(S, N, device=dev)
        # two calls: sampling state (acc, counts, step0) must carry over between calls
        idx = pop.forward(20, x=None if x is None else x[:20].to(dev), w_in=None if w_in is None else w_in.to(dev),
                          cutoff=5, sampling_steps=3, samples=samples, rec={0: rec}, out_var=_abi.RP_OUT_STATE0)
        idx = pop.forward(T - 20, x=None if x is None else x[20:].to(dev),
                          w_in=None if w_in is None else w_in.to(dev), step0=20, cutoff=5, sampling_steps=3,
                          samples=samples, rec={0: rec}, sample_idx=idx, out_var=_abi.RP_OUT_STATE0)
        res.append((idx, samples, rec, pop.v))
    assert res[0][0] == res[1][0] == len([t for t in range(T) if t >= 5 and t % 3 == 0])
    for a, b, name in zip(res[0][1:], res[1][1:], ("samples", "rec", "v")):
        assert_same(a, b, f"{mode} {name}")


@pytest.mark.parametrize("N", [32, 64, 1024])
def test_qif_forward_spikes_bit_exact(oracle_backend, hip_device, N):
    T = 3000
    gen = torch.Generator().manual_seed(N)
    W = torch.rand(N, N, generator=gen) * 20.0 / N
    params = dict(tau=1.0, k=1.0, eta=-5.0 + torch.randn(N, generator=gen), tau_s=1.0)
    x = torch.full((T, 1), 15.0)
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        pop = Population("qif", W, params, 1e-3, dev, v0=-2.0, s0=0.0, theta=100.0, v_reset=-100.0, backend=backend)
        ev = torch.zeros(200000, 2, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        traj = torch.zeros(T, N, device=dev)
        pop.forward(T, x=x.to(dev), traj=traj, out_var=_abi.RP_OUT_SOURCE, spike_events=ev, spike_count=cnt)
        c = int(cnt.item())
        e = ev[:c].cpu().numpy()
        e = e[np.lexsort((e[:, 1], e[:, 0]))]
        res.append((c, e, traj, pop.v, pop.source))
    assert res[0][0] == res[1][0] and res[0][0] > N, "spike counts differ or network silent"
    assert (res[0][1] == res[1][1]).all(), "spike (step, index) events differ"
    for a, b, name in zip(res[0][2:], res[1][2:], ("traj", "v", "s")):
        assert_same(a, b, f"qif N={N} {name}")


def test_sharded_rows_match_unsharded(hip_device):
    """Row shards on one device reproduce the unsharded result bit for bit (the summation tree
    does not depend on the sharding)."""
    N, T, P = 1024, 25, 4
    W, params, v0 = make_li(N, seed=11)
    x = torch.randn(T, N, generator=torch.Generator().manual_seed(2)).to(hip_device)
    full = Population("li_tanh", W, params, 1e-2, hip_device, v0=v0)
    traj_full = torch.zeros(T, N, device=hip_device)
    full.forward(T, x=x, traj=traj_full, out_var=_abi.RP_OUT_SOURCE)
    rows = N // P
    shards = [Population("li_tanh", W[p * rows:(p + 1) * rows], {k: v[p * rows:(p + 1) * rows] for k, v in params.items()},
                         1e-2, hip_device, v0=v0[p * rows:(p + 1) * rows], n=N, row0=p * rows) for p in range(P)]
    r_full = torch.cat([s.source_local for s in shards])
    traj = torch.zeros(T, N, device=hip_device)
    for t in range(T):
        for s in shards:
            s.source.copy_(r_full)
        for p, s in enumerate(shards):
            s.forward(1, x=x[t:t + 1, p * rows:(p + 1) * rows], traj=traj[t:t + 1, p * rows:(p + 1) * rows],
                      out_var=_abi.RP_OUT_SOURCE)
        r_full = torch.cat([s.source_local for s in shards])  # the "all-gather"
    assert_same(traj_full, traj, "sharded trajectory")


@pytest.mark.parametrize("N,T", [(48, 20), (256, 16), (1000, 12), (1028, 9)])
@pytest.mark.parametrize("model,grad_var", [("li_tanh", _abi.RP_OUT_STATE0), ("li_tanh", _abi.RP_OUT_SOURCE),
                                            ("li_sigmoid", _abi.RP_OUT_SOURCE)])
@pytest.mark.parametrize("transposed", [False, True])
def test_rate_backward_bit_exact(oracle_backend, hip_device, N, T, model, grad_var, transposed):
    """Adjoint sweep, both layouts -- column-strip W^T u + finalize (large n) and the fused row-dot launch on the
    transposed copy (rp_bwd.wt, small n) -- with parameter gradients: HIP == oracle, bit for bit."""
    W, params, v0 = make_li(N, seed=N + 7, model=model)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(T, N, generator=gen) * 0.5
    g_out = torch.randn(T, N, generator=gen) * 0.1
    lam_T = torch.randn(N, generator=gen) * 0.01
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        pop = Population(model, W, params, 1e-2, dev, v0=v0, backend=backend)
        pop.transposed_adjoint = transposed
        tv, tr, ta, ut, gx = (torch.zeros(T, N, device=dev) for _ in range(5))
        pop.forward(T, x=x.to(dev), save_v=tv, save_src=tr, save_acc=ta)
        lam = lam_T.clone().to(dev)
        gt, gk, ge = (torch.zeros(N, device=dev) for _ in range(3))
        # the sigmoid's own parameters (r_max, s, v0) have gradient kernels too
        ga = {q: torch.zeros(N, device=dev) for q in (("r_max", "s", "v0") if model == "li_sigmoid" else ())}
        keep = pop.backward(tv, tr, g_out.to(dev), grad_var, lam, ut, traj_acc=ta, grad_tau=gt, grad_k=gk,
                            grad_eta=ge, grad_x=gx, grad_params=ga)
        if dev != "cpu":
            torch.cuda.synchronize()
        res.append((lam, ut, gt, gk, ge, gx) + tuple(ga.values()))
        del keep
    assert model != "li_sigmoid" or all(float(t.abs().max()) > 0 for t in res[0][6:])
    for a, b, name in zip(res[0], res[1], ("lambda_0", "u_traj", "grad_tau", "grad_k", "grad_eta", "grad_x", "grad_r_max",
                                           "grad_s", "grad_v0")):
        assert_same(a, b, f"{model} N={N} {name}")
    assert float(res[0][0].abs().max()) > 0


@pytest.mark.parametrize("N,T", [(48, 7), (128, 16), (200, 100), (260, 33),
                                 # full 128 x 128 tiles: the LDS-DMA ring kernel -- one slice, exactly three slices (the ring
                                 # depth), many slices + a ragged last one, a single ragged slice, T = 1
                                 (256, 100), (384, 48), (128, 33), (640, 5), (256, 1), (128, 160),
                                 # T % 16 == 1: the ragged slice holds ONE row, so three of the four waves issue no load for
                                 # it (the counted wait in front of the slice before it must not rely on them)
                                 (256, 17), (128, 49), (384, 161)])
def test_outer_accum_bit_exact(oracle_backend, hip_device, N, T):
    """MFMA f32 dW GEMM == t-ordered fmaf chain of the oracle, bit for bit (incl. ragged tiles; both kernels: register-
    staged slices for ragged shapes, the LDS-DMA ring on full tiles)."""
    gen = torch.Generator().manual_seed(N + T)
    A = torch.randn(T, N, generator=gen)
    B = torch.randn(T, N, generator=gen)
    C0 = torch.randn(N, N, generator=gen)
    # asymmetric operands: a swapped row/col map in the C write-back cannot pass
    A[:, 0] += 3.0
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        pop = Population("li_tanh", torch.zeros(N, N), dict(tau=1.0, k=1.0, eta=0.0), 1e-2, dev, backend=backend)
        Cm = C0.clone().to(dev)
        pop.outer_accum(0.37, A.to(dev), B.to(dev), Cm)
        res.append(Cm)
    assert_same(res[0], res[1], f"outer_accum N={N} T={T}")
    ref = C0.double() + 0.37 * (A.double().T @ B.double())
    assert (res[1].cpu().double() - ref).abs().max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("N,T", [(256, 37), (130, 9), (384, 16), (77, 5)])
def test_outer_product_store_form_never_reads_its_result(oracle_backend, hip_device, N, T):
    """rp_outer_store_rect (what the autograd windows call for dW): C = alpha A^T B written into memory full of NaN -- a
    read of C anywhere would poison the result -- with the bits of rp_outer_accum into zeros, on both GEMM kernels."""
    from rectipy_amd.engine import HipBackend
    gen = torch.Generator().manual_seed(N * T)
    A = torch.randn(T, N, generator=gen)
    B = torch.randn(T, N, generator=gen)
    A[:, 0] += 3.0
    pop = Population("li_tanh", torch.zeros(N, N), dict(tau=1.0, k=1.0, eta=0.0), 1e-2, hip_device)
    a, b = A.to(hip_device), B.to(hip_device)
    acc = torch.zeros(N, N, device=hip_device)
    pop.outer_accum(0.37, a, b, acc)
    out = torch.full((N, N), float("nan"), device=hip_device)
    HipBackend().outer_store_rect(N, N, T, 0.37, a.data_ptr(), N, b.data_ptr(), N, out.data_ptr(), N, torch.device(hip_device))
    torch.cuda.synchronize()
    assert_same(acc, out, f"store form vs accumulate into zeros N={N} T={T}")
    assert_same(acc, pop.outer_product(0.37, a, b), "Population.outer_product")
    cpu = Population("li_tanh", torch.zeros(N, N), dict(tau=1.0, k=1.0, eta=0.0), 1e-2, "cpu", backend=oracle_backend)
    assert_same(cpu.outer_product(0.37, A, B), out, "oracle mirror of the store form")


def test_full_size_properties(hip_device):
    """BASELINE size (N = 32768): properties that do not need the oracle to run at that size.
    (1) sampled rows of the fused step agree with a float64 dot product;  (2) the transposed product of the
    adjoint kernel agrees with float64 W^T u (so forward and backward read the same matrix consistently);
    (3) row-sharding the same step over 4 shards is bit-identical."""
    N = 32768
    dev = hip_device
    gen = torch.Generator(device=dev).manual_seed(0)
    W = torch.randn(N, N, device=dev, generator=gen) * (1.5 / N ** 0.5)
    v0 = torch.randn(N, device=dev, generator=gen)
    tau = torch.rand(N, device=dev, generator=gen) * 5 + 5
    k = torch.rand(N, device=dev, generator=gen) + 0.5
    params = dict(tau=tau, k=k, eta=0.0)
    pop = Population("li_tanh", W, params, 1e-2, dev, v0=v0)
    r0 = pop.source.clone()
    acc = torch.zeros(1, N, device=dev)
    tv = torch.zeros(1, N, device=dev)
    tr = torch.zeros(1, N, device=dev)
    pop.forward(1, save_acc=acc, save_v=tv, save_src=tr)
    assert torch.equal(tr[0], r0) and torch.equal(tv[0], v0)
    rows = torch.randint(0, N, (64,), generator=torch.Generator().manual_seed(1)).to(dev)
    ref = W[rows].double() @ r0.double()
    assert (acc[0, rows].double() - ref).abs().max() < 2e-5
    v_ref = v0.double() + 1e-2 * (-v0.double() / tau.double() + k.double() * (W.double() @ r0.double()))
    assert (pop.v.double() - v_ref).abs().max() < 2e-5
    # (2) adjoint of the same step
    lam_T = torch.randn(N, device=dev, generator=gen)
    lam = lam_T.clone()
    ut = torch.zeros(1, N, device=dev)
    keep = pop.backward(tv, tr, None, _abi.RP_OUT_STATE0, lam, ut)
    u = (k * lam_T).double()
    z = W.double().T @ u
    lam_ref = lam_T.double() + 1e-2 * (-lam_T.double() / tau.double() + (1 - r0.double() ** 2) * z)
    assert (lam.double() - lam_ref).abs().max() < 5e-5
    del keep, z, v_ref
    # (3) sharding invariance at full size
    P = 4
    shard_rows = N // P
    v1 = pop.v.clone()
    for p in range(P):
        sl = slice(p * shard_rows, (p + 1) * shard_rows)
        sh = Population("li_tanh", W[sl], dict(tau=tau[sl], k=k[sl], eta=0.0), 1e-2, dev, v0=v0[sl], n=N,
                        row0=p * shard_rows)
        sh.source.copy_(r0)
        sh.forward(1)
        assert torch.equal(sh.v, v1[sl])


def test_full_size_bptt_window(hip_device):
    """BASELINE configs[2] size (N = 32768): a T = 8 BPTT window with a non-zero per-step output gradient G_t.
    The column-strip adjoint (state, dL/dtau, dL/dk, dL/deta) and the 65 536-tile MFMA weight-gradient GEMM (XCD-swizzled
    launch) are compared with a float64 recomputation on the device from the SAME stored trajectory:
      u_t = k * lam_{t+1};  z = W^T u_t;  lam_t = lam_{t+1} + dt (-lam_{t+1}/tau + (1 - r_t^2) z) + G_t;
      dL/deta += dt lam_{t+1};  dL/dk += dt lam_{t+1} (W r_t);  dL/dtau += dt lam_{t+1} v_t / tau^2;  dW = dt sum_t u_t r_t^T.
    Tolerances (fp32 sums of 32768 terms vs float64): 5e-5 of each quantity's max magnitude; GEMM entries 1e-5 of max."""
    N, T, dt, dev = 32768, 8, 1e-2, hip_device
    gen = torch.Generator(device=dev).manual_seed(10)
    W = torch.randn(N, N, device=dev, generator=gen) * (1.5 / N ** 0.5)
    v0 = torch.randn(N, device=dev, generator=gen)
    tau = torch.rand(N, device=dev, generator=gen) * 5 + 5
    k = torch.rand(N, device=dev, generator=gen) + 0.5
    eta = torch.randn(N, device=dev, generator=gen) * 0.1
    x = torch.randn(T, N, device=dev, generator=gen) * 0.5
    G = torch.randn(T, N, device=dev, generator=gen) * 0.01
    lam_T = torch.randn(N, device=dev, generator=gen) * 0.01
    pop = Population("li_tanh", W, dict(tau=tau, k=k, eta=eta), dt, dev, v0=v0)
    assert pop.n > _abi.RP_BWD_TRANSPOSE_MAX_N  # the column-strip kernels, not the transposed copy
    tv, tr, ta, ut = (torch.zeros(T, N, device=dev) for _ in range(4))
    pop.forward(T, x=x, save_v=tv, save_src=tr, save_acc=ta)
    lam = lam_T.clone()
    g_tau, g_k, g_eta = (torch.zeros(N, device=dev) for _ in range(3))
    keep = pop.backward(tv, tr, G, _abi.RP_OUT_STATE0, lam, ut, traj_acc=ta, grad_tau=g_tau, grad_k=g_k, grad_eta=g_eta)
    gW = torch.zeros(N, N, device=dev)
    pop.outer_accum(dt, ut, tr, gW)
    torch.cuda.synchronize()
    del keep

    # float64 recomputation from the stored trajectory
    d = torch.float64
    Wd = W.to(d)
    l = lam_T.to(d)
    r_tau, r_k, r_eta = (torch.zeros(N, dtype=d, device=dev) for _ in range(3))
    u_ref = torch.zeros(T, N, dtype=d, device=dev)
    for t in range(T - 1, -1, -1):
        r_t, v_t = tr[t].to(d), tv[t].to(d)
        u = k.to(d) * l
        u_ref[t] = u
        r_eta += dt * l
        r_k += dt * l * (Wd @ r_t)
        r_tau += dt * l * v_t / tau.to(d) ** 2
        z = Wd.T @ u
        l = l + dt * (-l / tau.to(d) + (1 - r_t ** 2) * z) + G[t].to(d)

    def rel(a, b):
        return float((a.to(d) - b).abs().max() / b.abs().max())

    assert rel(lam, l) < 5e-5, f"adjoint state: {rel(lam, l)}"
    assert rel(ut, u_ref) < 5e-5
    assert rel(g_eta, r_eta) < 5e-5 and rel(g_k, r_k) < 5e-5 and rel(g_tau, r_tau) < 5e-5
    # the stored W r_t itself (forward) against float64, sampled rows
    rows = torch.randint(0, N, (64,), generator=torch.Generator().manual_seed(11)).to(dev)
    assert ((Wd[rows] @ tr[3].to(d)) - ta[3, rows].to(d)).abs().max() < 2e-5
    # dW GEMM: 64 x 64 sampled entries spread over the tile grid + two whole rows / columns (edges of the swizzled grid)
    cols = torch.randint(0, N, (64,), generator=torch.Generator().manual_seed(12)).to(dev)
    ref_blk = dt * (u_ref[:, rows].T @ tr[:, cols].to(d))
    scale = float(ref_blk.abs().max())
    assert float((gW[rows][:, cols].to(d) - ref_blk).abs().max()) < 1e-5 * scale
    for i in (0, N - 1, 12345):
        assert float((gW[i].to(d) - dt * (u_ref[:, i] @ tr.to(d))).abs().max()) < 1e-5 * scale
        assert float((gW[:, i].to(d) - dt * (u_ref.T @ tr[:, i].to(d))).abs().max()) < 1e-5 * scale
    # whole-matrix check: every one of the 256 x 256 tiles contributes to these two projections
    ones = torch.ones(N, dtype=d, device=dev)
    full_scale = float((dt * (u_ref.abs().T @ (tr.to(d).abs() @ ones))).max())
    assert float(((gW.to(d) @ ones) - dt * (u_ref.T @ (tr.to(d) @ ones))).abs().max()) < 2e-6 * full_scale
    assert float(((ones @ gW.to(d)) - dt * ((ones @ u_ref.T) @ tr.to(d))).abs().max()) < 2e-6 * full_scale


def test_config5_shard_shape_one_step(hip_device):
    """BASELINE configs[4]: the per-GPU shard of the N = 262144 network (32768 rows x 262144 columns, 34.4 GB) at a
    non-zero shard offset.  One step from a gathered source vector: sampled rows of W_p r against a float64 dot
    product, the Euler update of the shard's neurons, the shard's slice of the next source vector (and nothing
    else of it) written.  The staggered start `m0` depends on the GLOBAL row index, so the offset matters."""
    rows, n, dt, dev = 32768, 262144, 1e-2, hip_device
    row0 = 5 * rows
    gen = torch.Generator(device=dev).manual_seed(20)
    W = torch.empty(rows, n, device=dev)
    for a in range(0, rows, 4096):  # generated slab by slab: no 34 GB temporary next to the matrix
        W[a:a + 4096] = torch.randn(4096, n, device=dev, generator=gen) * (1.5 / n ** 0.5)
    v0 = torch.randn(rows, device=dev, generator=gen)
    tau = torch.rand(rows, device=dev, generator=gen) * 5 + 5
    k = torch.rand(rows, device=dev, generator=gen) + 0.5
    r_full = torch.tanh(torch.randn(n, device=dev, generator=gen))
    pop = Population("li_tanh", W, dict(tau=tau, k=k, eta=0.0), dt, dev, v0=v0, n=n, row0=row0)
    pop.source.copy_(r_full)  # what the all-gather of the previous step delivers
    nxt = pop.src[pop.cur ^ 1]
    nxt.fill_(7.0)
    acc = torch.zeros(1, rows, device=dev)
    pop.forward(1, save_acc=acc)
    pick = torch.cat([torch.tensor([0, 7, 8, rows - 1]),
                      torch.randint(0, rows, (60,), generator=torch.Generator().manual_seed(21))]).to(dev)
    ref = W[pick].double() @ r_full.double()
    assert (acc[0, pick].double() - ref).abs().max() < 5e-5
    v_ref = v0[pick].double() + dt * (-v0[pick].double() / tau[pick].double() + k[pick].double() * ref)
    assert (pop.v[pick].double() - v_ref).abs().max() < 5e-5
    assert pop.cur == 1 or pop.source is nxt
    assert torch.allclose(nxt[row0:row0 + rows], torch.tanh(pop.v), atol=2e-6)
    assert bool((nxt[:row0] == 7.0).all()) and bool((nxt[row0 + rows:] == 7.0).all())
    # the same rows as an UNSHARDED-order check: a shard at another offset with the same data gives the same sums
    # only when the global row index matches -- rows 0..7 moved by a multiple of the stagger period reproduce bit for bit
    M = n // 256
    shift = 8 * M  # 3 * ((row0 + shift) >> 3) == 3 * (row0 >> 3) (mod M)
    if row0 + shift + rows <= n:
        pop2 = Population("li_tanh", W, dict(tau=tau, k=k, eta=0.0), dt, dev, v0=v0, n=n, row0=row0 + shift)
        pop2.source.copy_(r_full)
        acc2 = torch.zeros(1, rows, device=dev)
        pop2.forward(1, save_acc=acc2)
        assert torch.equal(acc, acc2)


def test_qif_full_size_uncoupled_neurons_match_scalar_model(hip_device):
    """N = 65536 QIF population with W = 0 in effect (k = 0): every neuron is an independent scalar QIF unit, so
    a handful of neurons can be re-simulated in float32 numpy with the reference's update order and must match
    exactly, spike steps included."""
    N, T = 65536, 2000
    dev = hip_device
    W = torch.zeros(N, N, device=dev)
    eta = torch.linspace(-6.0, 4.0, N, device=dev)
    pop = Population("qif", W, dict(tau=1.0, k=0.0, eta=eta, tau_s=1.0), 1e-3, dev, v0=-2.0, s0=0.0, theta=100.0,
                     v_reset=-100.0)
    ev = torch.zeros(4_000_000, 2, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    x = torch.full((T, 1), 15.0, device=dev)
    pop.forward(T, x=x, spike_events=ev, spike_count=cnt)
    c = int(cnt.item())
    assert 0 < c <= ev.shape[0]
    events = ev[:c].cpu().numpy()
    f = np.float32
    dt, inv_dt = f(1e-3), f(1.0) / f(1e-3)
    for i in (0, 1234, 40000, 65535):
        v, s, e, spikes = f(-2.0), f(0.0), eta[i].cpu().numpy().astype(np.float32), []
        for t in range(T):
            sp = f(1.0) if v - f(100.0) >= 0 else f(0.0)
            dv = f(f(f(v * v) + e) + f(15.0)) / f(1.0) + f(0.0)
            ds = f(-s / f(1.0)) + f(sp * inv_dt)
            v_new = f(v + f(dt * dv))
            s = f(s + f(dt * ds))
            v = f(f(v_new * f(1.0 - sp)) + f(sp * f(-100.0)))
            if sp:
                spikes.append(t)
        got = sorted(events[events[:, 1] == i][:, 0].tolist())
        assert got == spikes, f"neuron {i}: spike steps differ"
        assert float(pop.v[i]) == float(v) and float(pop.source[i]) == float(s)


def test_qif_full_size_dense_step_properties(hip_device):
    """BASELINE config 4 size (N = 65536, dense coupling): one step from a random state.  Sampled neurons are
    recomputed from a float64 dot product; spiking neurons are exactly those with v >= theta, are reset to
    v_reset, and their s jumps by exactly 1 relative to a non-spiking twin."""
    N, dev = 65536, hip_device
    gen = torch.Generator(device=dev).manual_seed(4)
    W = torch.rand(N, N, device=dev, generator=gen) * (20.0 / N)
    v0 = torch.rand(N, device=dev, generator=gen) * 220.0 - 110.0  # some above theta = 100
    v0[7] = 100.0  # equality spikes
    s0 = torch.rand(N, device=dev, generator=gen)
    eta = torch.randn(N, device=dev, generator=gen) - 5.0
    pop = Population("qif", W, dict(tau=1.0, k=1.0, eta=eta, tau_s=1.0), 1e-3, dev, v0=v0, s0=s0, theta=100.0,
                     v_reset=-100.0)
    ev = torch.zeros(N, 2, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    acc = torch.zeros(1, N, device=dev)
    pop.forward(1, x=torch.full((1, 1), 15.0, device=dev), save_acc=acc, spike_events=ev, spike_count=cnt, step0=5)
    spiking = v0 >= 100.0
    c = int(cnt.item())
    assert c == int(spiking.sum()) and bool(spiking[7])
    events = ev[:c].cpu().numpy()
    assert (events[:, 0] == 5).all()
    assert sorted(events[:, 1].tolist()) == torch.nonzero(spiking).flatten().cpu().tolist()
    rows = torch.randint(0, N, (64,), generator=torch.Generator().manual_seed(1)).to(dev)
    ref_acc = W[rows].double() @ s0.double()
    assert (acc[0, rows].double() - ref_acc).abs().max() < 2e-5
    v_new, s_new = pop.v, pop.source
    assert bool((v_new[spiking] == -100.0).all())
    quiet = ~spiking
    v_ref = v0.double() + 1e-3 * ((v0.double() ** 2 + eta.double() + 15.0) + W[rows].double().sum() * 0)
    dv = (v0[rows].double() ** 2 + eta[rows].double() + 15.0) + ref_acc
    ok = quiet[rows]
    assert ((v_new[rows].double() - (v0[rows].double() + 1e-3 * dv))[ok]).abs().max() < 2e-4
    s_ref = s0.double() * (1 - 1e-3) + spiking.double()
    assert (s_new.double() - s_ref).abs().max() < 1e-5


def test_run_to_run_determinism(hip_device, measure_backend):
    """No atomics on the numerical path and a fixed summation tree: two runs of the same window are bit-identical
    (forward trajectory, adjoint state, weight gradient), and so is a different launch configuration (measurement build)."""
    N, T = 4096, 300
    W, params, v0 = make_li(N, seed=5)
    x = torch.randn(T, 1, generator=torch.Generator().manual_seed(6)).to(hip_device)
    g_out = (torch.randn(T, N, generator=torch.Generator().manual_seed(7)) * 0.01).to(hip_device)
    runs = []
    for tune in ((0, 0), (0, 0), (2, 4)):
        measure_backend.lib.rp_set_tuning(tune[0], tune[1], 1, 0)
        pop = Population("li_tanh", W, params, 1e-2, hip_device, v0=v0, backend=measure_backend if tune[0] else None)
        tv, tr, ut = (torch.zeros(T, N, device=hip_device) for _ in range(3))
        pop.forward(T, x=x, save_v=tv, save_src=tr)
        lam = torch.zeros(N, device=hip_device)
        keep = pop.backward(tv, tr, g_out, _abi.RP_OUT_STATE0, lam, ut)
        gW = torch.zeros(N, N, device=hip_device)
        pop.outer_accum(pop.dt, ut, tr, gW)
        torch.cuda.synchronize()
        runs.append((tv.clone(), lam.clone(), gW))
        del keep
    measure_backend.lib.rp_set_tuning(0, 0, 1, 0)
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)


def test_sharded_adjoint_steps_and_rect_gemm_bit_exact(oracle_backend, hip_device):
    """rp_bwd_prepare / rp_bwd_partial / rp_bwd_apply on one row shard and the rectangular dW GEMM: HIP == oracle.
    (The reduce-scatter between partial and apply is emulated by taking the shard's own slice.)"""
    from rectipy_amd import _abi as A
    N, T, P, p = 1024, 14, 4, 1
    rows, row0 = N // P, (N // P) * p
    W, params, v0 = make_li(N, seed=77)
    gen = torch.Generator().manual_seed(78)
    x = torch.randn(T, rows, generator=gen) * 0.5
    g_out = torch.randn(T, rows, generator=gen) * 0.1
    r_full = torch.tanh(torch.randn(T + 1, N, generator=gen))
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        pop = Population("li_tanh", W[row0:row0 + rows], {k: v[row0:row0 + rows] for k, v in params.items()}, 1e-2, dev,
                         v0=v0[row0:row0 + rows], n=N, row0=row0, backend=backend)
        be = pop.backend
        tv, tr, ta, ut, gx = (torch.zeros(T, rows, device=dev) for _ in range(5))
        for t in range(T):  # the other shards' source slices come from outside (all-gather)
            pop.source.copy_(r_full[t].to(dev))
            pop.forward(1, x=x[t:t + 1].to(dev), save_v=tv[t:t + 1], save_src=tr[t:t + 1], save_acc=ta[t:t + 1])
        lam = torch.zeros(rows, device=dev)
        gt, gk, ge = (torch.zeros(rows, device=dev) for _ in range(3))
        partials = torch.zeros(be.shard_partial_rows(N, rows), N, device=dev)
        z_full = torch.zeros(N, device=dev)
        go = g_out.to(dev).contiguous()
        b = A.RpBwd()
        b.steps, b.ld, b.ldg, b.ldgx, b.grad_var = T, rows, rows, rows, A.RP_OUT_SOURCE
        b.traj_v, b.traj_r, b.traj_acc = tv.data_ptr(), tr.data_ptr(), ta.data_ptr()
        b.grad_out, b.lam, b.u_traj, b.partials = go.data_ptr(), lam.data_ptr(), ut.data_ptr(), partials.data_ptr()
        b.grad_tau, b.grad_k, b.grad_eta, b.grad_x = gt.data_ptr(), gk.data_ptr(), ge.data_ptr(), gx.data_ptr()
        net = pop._net()
        be.bwd_prepare(net, b, pop.device)
        zs = []
        for t in range(T - 1, -1, -1):
            be.bwd_partial(net, b, t, z_full.data_ptr(), pop.device)
            zs.append(z_full.clone())
            z_local = z_full[row0:row0 + rows].clone()
            be.bwd_apply(net, b, t, z_local.data_ptr(), pop.device)
        gW = torch.zeros(rows, N, device=dev)
        rf = r_full[:T].to(dev).contiguous()
        be.outer_accum_rect(rows, N, T, 1e-2, ut.data_ptr(), rows, rf.data_ptr(), N, gW.data_ptr(), N, pop.device)
        if dev != "cpu":
            torch.cuda.synchronize()
        res.append((torch.stack(zs), lam, ut, gt, gk, ge, gx, gW))
    for a, b_, name in zip(res[0], res[1], ("z_full", "lambda", "u_traj", "grad_tau", "grad_k", "grad_eta", "grad_x", "dW")):
        assert_same(a, b_, f"sharded {name}")
    assert float(res[0][7].abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["qif", "ik"])
def test_sharded_spiking_adjoint_bit_exact(oracle_backend, hip_device, model):
    """The step-wise sharded adjoint of the spiking models (rp_bwd_prepare / partial / apply dispatching to the spiking
    kernels) on one row shard: HIP == oracle, bit for bit.  The reduce-scatter is emulated by the shard's own slice."""
    N, T, Pn, p = 1024, 20, 4, 2
    rows, row0 = N // Pn, (N // Pn) * p
    gen = torch.Generator().manual_seed(91)
    rnd = lambda lo, hi, n=rows: torch.rand(n, generator=gen) * (hi - lo) + lo
    W = torch.rand(rows, N, generator=gen) * 10.0 / N
    if model == "qif":
        params = dict(tau=rnd(0.8, 1.2), k=rnd(0.8, 1.2), eta=rnd(-6, -4), tau_s=rnd(0.5, 1.5))
        kw = dict(v0=rnd(-5, 90), s0=rnd(0, 1), theta=100.0, v_reset=-100.0)
        dt, x, names = 1e-3, 15.0 + torch.randn(T, 1, generator=gen), ("tau", "k", "eta", "tau_s")
    else:
        params = dict(C=rnd(90, 110), k=rnd(0.6, 0.8), v_r=-60.0, v_theta=-40.0, eta=rnd(-5, 5), g=rnd(0.8, 1.2), E_r=0.0,
                      b=rnd(-2.5, -1.5), tau_u=rnd(30, 36), kappa=rnd(8, 12), tau_s=rnd(5, 7))
        kw = dict(v0=rnd(-60, 38), s0=rnd(0, 0.5), aux0=rnd(0, 20), theta=40.0, v_reset=-55.0)
        dt, x = 1e-1, 120.0 + 5 * torch.randn(T, rows, generator=gen)
        names = ("C", "k", "eta", "tau_s", "g", "b", "tau_u", "kappa", "v_r", "v_theta", "E_r")
    s_full = torch.rand(T + 1, N, generator=gen) * 0.5
    g_out = torch.randn(T, rows, generator=gen) * 0.1
    l0 = [torch.randn(rows, generator=gen) * 0.01 for _ in range(3)]

    class OwnSlice:  # stands in for the cross-rank reduce-scatter
        comm = None

        @staticmethod
        def reduce_scatter(z_full, z_local):
            z_local.copy_(z_full[row0:row0 + rows])

    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        pop = Population(model, W, params, dt, dev, n=N, row0=row0, backend=backend, **kw)
        tv, ts, ta, tx, ut, gx = (torch.zeros(T, rows, device=dev) for _ in range(6))
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        ev = torch.zeros(rows * T, 2, dtype=torch.int32, device=dev)
        for t in range(T):  # the other shards' slices of s come from outside (all-gather)
            own = pop.source_local.clone()
            pop.source.copy_(s_full[t].to(dev))
            pop.set_source_local(own)
            pop.forward(1, x=x[t:t + 1].to(dev), save_v=tv[t:t + 1], save_src=ts[t:t + 1], save_acc=ta[t:t + 1],
                        save_aux=tx[t:t + 1] if pop.aux is not None else None, spike_events=ev, spike_count=cnt, step0=t)
        lv, ls, lx = (t_.clone().to(dev) for t_ in l0)
        grads = {q: torch.zeros(rows, device=dev) for q in names}
        keep = pop.backward(tv, ts, g_out.to(dev), _abi.RP_OUT_SOURCE, lv, ut, traj_acc=ta, grad_x=gx, lam_s=ls, slope=0.7,
                            x=x.to(dev), traj_aux=tx if pop.aux is not None else None,
                            lam_x=lx if pop.aux is not None else None, grad_params=grads, exchange=OwnSlice)
        if dev != "cpu":
            torch.cuda.synchronize()
        res.append(([lv, ls, lx, ut, gx] + [grads[q] for q in names], int(cnt.item())))
        del keep
    assert res[0][1] == res[1][1] and res[0][1] > 0, "the window must contain spikes"
    for a, b_, name in zip(res[0][0], res[1][0], ["lambda_v", "lambda_s", "lambda_aux", "u_traj", "grad_x"] + list(names)):
        assert_same(a, b_, f"sharded {model} {name}")


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,T", [(1024, 8192, 6), (4096, 4096, 5)])
def test_outer_accum_xcd_tile_order_bit_exact(oracle_backend, hip_device, m, n, T):
    """Sizes at which the GEMM uses the XCD-aware tile order (8 x 8 super-tiles dealt to the 8 XCDs): which workgroup
    computes which tile must not change a bit, and every tile must be computed exactly once."""
    gen = torch.Generator().manual_seed(m + n + T)
    A = torch.randn(T, m, generator=gen)
    B = torch.randn(T, n, generator=gen)
    C0 = torch.randn(m, n, generator=gen)
    res = []
    for be, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        if be is None:
            from rectipy_amd.engine import HipBackend
            be = HipBackend()
        a, b_, c = A.to(dev).contiguous(), B.to(dev).contiguous(), C0.clone().to(dev)
        be.outer_accum_rect(m, n, T, 0.37, a.data_ptr(), m, b_.data_ptr(), n, c.data_ptr(), n, torch.device(dev))
        if dev != "cpu":
            torch.cuda.synchronize()
        res.append(c)
    assert_same(res[0], res[1], f"outer_accum_rect {m}x{n} T={T}")


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["li_tanh", "qif"])
def test_long_run_stays_bit_exact(oracle_backend, hip_device, model):
    """3000 consecutive steps (chaotic regime for tanh, thousands of spikes for QIF): the HIP state, the sampled
    output and the spike events equal the oracle's bit for bit at the end -- no drift, no order-dependent sums."""
    N, T = 520, 3000
    gen = torch.Generator().manual_seed(12)
    if model == "li_tanh":
        W, params, v0 = make_li(N, seed=5)
        kw, dt = dict(v0=v0), 1e-2
        x = torch.sin(torch.arange(T) * 0.01).reshape(-1, 1)
    else:
        W = torch.rand(N, N, generator=gen) * 20.0 / N
        params = dict(tau=1.0, k=1.0, eta=-5.0 + torch.randn(N, generator=gen), tau_s=1.0)
        kw, dt = dict(v0=-2.0, s0=0.0, theta=100.0, v_reset=-100.0), 1e-3
        x = torch.full((T, 1), 15.0)
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        pop = Population(model, W, params, dt, dev, backend=backend, **kw)
        samples = torch.zeros(T // 50 + 2, N, device=dev)
        ev = torch.zeros(1 << 16, 2, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        extra = dict(spike_events=ev, spike_count=cnt) if model == "qif" else {}
        n_s = pop.forward(T, x=x.to(dev), out_var=_abi.RP_OUT_SOURCE, sampling_steps=50, samples=samples, **extra)
        if dev != "cpu":
            torch.cuda.synchronize()
        c = int(cnt.item())
        events = ev[:c].cpu().numpy()
        events = events[np.lexsort((events[:, 1], events[:, 0]))]  # the GPU appends a step's spikes in any order
        res.append((pop.v.cpu(), pop.source.cpu(), samples[:n_s].cpu(), events))
    for a, b_, name in zip(res[0][:3], res[1][:3], ("v", "source", "samples")):
        assert_same(a, b_, f"{model} after {T} steps: {name}")
    assert np.array_equal(res[0][3], res[1][3])
    if model == "qif":
        assert len(res[0][3]) > 500, "expected an actively spiking network"
    else:
        assert float(res[0][0].abs().max()) > 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("rows,n,row0", [(8192, 16384, 4096), (8200, 12300, 0)], ids=["aligned", "ragged"])
def test_measured_streaming_configuration_keeps_the_bits(rows, n, row0):
    """rp_tune_forward picks how W is walked (rows per wave, chunks in flight, resident workgroups) by timing; the
    summation order does not depend on it, so a run before and after the measurement gives identical bits, for the model
    that was timed (tanh) and for another one (QIF) that reuses the measurement."""
    import ctypes as C

    from rectipy_amd import _lib
    from rectipy_amd.engine import Population
    lib = _lib.load()
    dev = torch.device("cuda:0")
    assert rows * n * 4 > (320 << 20)                        # streams from HBM: the measurement runs
    gen = torch.Generator(device=dev).manual_seed(5)
    W = torch.randn(rows, n, device=dev, generator=gen) * (1.2 / n ** 0.5)
    x = torch.sin(torch.arange(12, device=dev) * 0.1).reshape(-1, 1).contiguous()
    os.environ["RECTIPY_AMD_NO_AUTOTUNE"] = "1"
    try:
        lib.rp_clear_forward_tuning()
        outs = {}
        for tuned in (False, True):
            pop = Population("li_tanh", W, dict(tau=2.0, k=1.0, eta=0.1), 1e-2, dev, n=n, row0=row0)
            pop.source.copy_(torch.tanh(torch.randn(n, device=dev, generator=torch.Generator(device=dev).manual_seed(6))))
            if tuned:
                w_before = W.clone()
                _lib.check(lib.rp_tune_forward(C.byref(pop._net()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
                assert torch.equal(W, w_before)
                info = pop.backend.forward_tuning(pop._net())
                assert info["measured"] and info["us_per_step"] > 0
                assert info["rows_per_wave"] in (1, 2, 4, 8) and info["workgroups"] >= 0
            else:
                assert not pop.backend.forward_tuning(pop._net())["measured"]
            traj = torch.empty(12, rows, device=dev)
            pop.forward(12, x=x, traj=traj)
            qif = Population("qif", W, dict(tau=1.0, k=1.0, eta=-5.0, tau_s=1.0), 1e-3, dev, n=n, row0=row0,
                             v0=torch.linspace(-2, 99, rows))
            qtraj = torch.empty(6, rows, device=dev)
            qif.forward(6, traj=qtraj)
            outs[tuned] = (traj, pop.v.clone(), qtraj, qif.v.clone())
        for a, b in zip(outs[False], outs[True]):
            assert torch.equal(a, b)
    finally:
        os.environ.pop("RECTIPY_AMD_NO_AUTOTUNE", None)
        lib.rp_clear_forward_tuning()


@pytest.mark.gpu
def test_measurement_policy_default_static_request_measures_cache_installs(tmp_path, monkeypatch):
    """rectipy_amd/tuning.py on the device: by default a population's first (short) forward runs on the static rule -- no
    probe launches; `tune=True` measures and persists the winner; the next population of that shape (a "second
    process": library table cleared) installs it from the cache file without measuring; every variant gives the same bits."""
    import time

    from rectipy_amd import _lib
    from rectipy_amd.engine import Population
    lib = _lib.load()
    dev = torch.device("cuda:0")
    rows, n = 8192, 16384
    monkeypatch.setenv("RECTIPY_AMD_TUNE_CACHE", str(tmp_path / "forward_tuning.json"))
    monkeypatch.delenv("RECTIPY_AMD_AUTOTUNE", raising=False)
    monkeypatch.delenv("RECTIPY_AMD_NO_AUTOTUNE", raising=False)
    lib.rp_clear_forward_tuning()
    gen = torch.Generator(device=dev).manual_seed(5)
    W = torch.randn(rows, n, device=dev, generator=gen) * (1.2 / n ** 0.5)
    x = torch.sin(torch.arange(12, device=dev) * 0.1).reshape(-1, 1).contiguous()

    def run(**kw):
        pop = Population("li_tanh", W, dict(tau=2.0, k=1.0, eta=0.1), 1e-2, dev, n=n, row0=4096, **kw)
        traj = torch.empty(12, rows, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pop.forward(12, x=x, traj=traj)
        torch.cuda.synchronize()
        return traj, pop.backend.forward_tuning(pop._net()), time.perf_counter() - t0

    try:
        run()                                                    # (first launch of the process: module load etc.)
        base, info, t_static = run()
        assert not info["measured"] and not (tmp_path / "forward_tuning.json").exists()
        tuned, info, t_tuned = run(tune=True)
        assert info["measured"] and (tmp_path / "forward_tuning.json").exists()
        assert t_tuned > 5 * t_static                            # the measurement is what a default run no longer pays
        lib.rp_clear_forward_tuning()                            # "second process"
        cached, info2, t_cached = run()
        assert info2["measured"] and (info2["rows_per_wave"], info2["chunks_in_flight"], info2["workgroups"]) == \
            (info["rows_per_wave"], info["chunks_in_flight"], info["workgroups"])
        assert t_cached < 0.2 * t_tuned
        assert torch.equal(base, tuned) and torch.equal(base, cached)
    finally:
        lib.rp_clear_forward_tuning()
