"""HIP kernels vs the CPU oracle through the C ABI -- bit-exact (same summation tree, same
roundings).  Sizes are small enough for the oracle to finish in seconds."""
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
