"""The resident step kernel (r05): all steps of an rp_forward call in ONE launch at launch-bound sizes (n <= 1024, n % 4 == 0) --
W in registers, the source vector handed from wave to wave as tagged 8-byte granules -- against one launch per step
(RECTIPY_AMD_RESIDENT=0): bit for bit, every model the per-step kernel integrates without population means, every run option."""
import numpy as np
import pytest
import torch

from rectipy_amd import _abi
from rectipy_amd.engine import Population

pytestmark = pytest.mark.gpu


def same(a, b, what):
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    eq = (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))
    assert eq.all(), f"{what}: {np.count_nonzero(~eq)} of {a.size} differ, max abs {np.abs(a - b).max()}"


def make(model, n, dev, seed=0):
    gen = torch.Generator().manual_seed(seed + n)
    u = lambda lo, hi: torch.rand(n, generator=gen) * (hi - lo) + lo
    if model in ("li_tanh", "li_sigmoid"):
        W = torch.randn(n, n, generator=gen) * 1.3 / n ** 0.5
        par = dict(tau=u(0.5, 2.0), k=u(0.5, 1.5), eta=torch.randn(n, generator=gen) * 0.1)
        if model == "li_sigmoid":
            par.update(r_max=u(0.8, 1.2), s=u(0.5, 2.0), v0=torch.randn(n, generator=gen) * 0.2)
        return Population(model, W, par, 1e-2, dev, v0=torch.randn(n, generator=gen) * 0.5)
    W = torch.rand(n, n, generator=gen) * 2.0 / n
    if model == "qif":
        par = dict(tau=u(0.8, 1.2), k=u(0.5, 1.5), eta=u(0.5, 3.0), tau_s=u(0.5, 2.0))
        return Population(model, W, par, 5e-3, dev, v0=torch.randn(n, generator=gen), theta=1.0, v_reset=-1.0)
    if model == "lif":
        par = dict(tau=u(0.4, 0.8), k=u(0.5, 1.5), eta=u(2.0, 5.0), tau_s=u(0.2, 0.6))
        return Population(model, W, par, 5e-3, dev, v0=torch.rand(n, generator=gen), theta=1.0, v_reset=0.0)
    raise AssertionError(model)


@pytest.mark.parametrize("model,n", [("li_tanh", 256), ("li_tanh", 512), ("li_tanh", 768), ("li_tanh", 1024), ("li_sigmoid", 512),
                                     ("qif", 512), ("qif", 1024), ("lif", 256),
                                     # ragged sizes: a partial last chunk, rows that do not fill the last wave, fewer than 256 columns
                                     ("li_tanh", 8), ("li_tanh", 100), ("li_tanh", 260), ("li_tanh", 500), ("li_tanh", 1000), ("li_sigmoid", 200),
                                     ("qif", 300), ("lif", 772),
                                     # n % 4 != 0: the 4-byte row-dot path (lane l: columns l, l + 64, ...) -- its own resident layout
                                     ("li_tanh", 5), ("li_tanh", 50), ("li_tanh", 125), ("li_tanh", 250), ("li_tanh", 999), ("li_sigmoid", 301),
                                     ("qif", 10), ("qif", 1023), ("lif", 150),
                                     # 1024 < n <= 1536 on the 16-byte path: five or six chunks, four rows per wave (wider: a launch per step either way)
                                     ("li_tanh", 1028), ("li_tanh", 1280), ("li_tanh", 1536), ("li_tanh", 1400), ("li_tanh", 2048),
                                     ("li_tanh", 1300), ("li_sigmoid", 1100), ("qif", 1536), ("qif", 1500), ("lif", 1284)])
def test_resident_launch_equals_a_launch_per_step(hip_device, monkeypatch, model, n):
    T = 57
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(T, n, generator=gen) * 0.5 + (1.5 if model in ("qif", "lif") else 0.0)).to(hip_device)
    got = []
    for resident in ("1", "0"):
        monkeypatch.setenv("RECTIPY_AMD_RESIDENT", resident)
        pop = make(model, n, hip_device)
        f32 = dict(dtype=torch.float32, device=hip_device)
        traj, sv, ss = torch.zeros(T, n, **f32), torch.zeros(T, n, **f32), torch.zeros(T, n, **f32)
        samples = torch.zeros(T, n, **f32)
        # two calls (the second starts from the first's state and sampling bookkeeping), cutoff and a sampling stride
        a = 23
        kw = dict(out_var=_abi.RP_OUT_SOURCE, cutoff=5, sampling_steps=3, samples=samples)
        i0 = pop.forward(a, x=x[:a], traj=traj[:a], save_v=sv[:a], save_src=ss[:a], step0=0, sample_idx=0, **kw)
        pop.forward(T - a, x=x[a:], traj=traj[a:], save_v=sv[a:], save_src=ss[a:], step0=a, sample_idx=i0, **kw)
        got.append(dict(traj=traj, save_v=sv, save_src=ss, samples=samples, acc=pop.acc.clone(), v=pop.v.clone(), src=pop.source.clone()))
    for key in got[0]:
        same(got[0][key], got[1][key], f"{model} n={n}: {key}")
    if model in ("qif", "lif"):
        assert float(got[0]["traj"].abs().max()) > 0   # (spikes happened: the synaptic variable moved)


def test_broadcast_and_projected_inputs(hip_device, monkeypatch):
    n, m, T = 512, 5, 40
    gen = torch.Generator().manual_seed(8)
    xb = torch.randn(T, 1, generator=gen).to(hip_device)
    xm = torch.randn(T, m, generator=gen).to(hip_device)
    w_in = torch.randn(n, m, generator=gen).to(hip_device)
    for kw in (dict(x=xb), dict(x=xm, w_in=w_in)):
        got = []
        for resident in ("1", "0"):
            monkeypatch.setenv("RECTIPY_AMD_RESIDENT", resident)
            pop = make("li_tanh", n, hip_device)
            traj = torch.zeros(T, n, device=hip_device)
            pop.forward(T, traj=traj, out_var=_abi.RP_OUT_STATE0, **kw)
            got.append((traj, pop.v.clone(), pop.source.clone()))
        for u, v, name in zip(got[0], got[1], ("traj", "v", "source")):
            same(u, v, f"{sorted(kw)}: {name}")


@pytest.mark.parametrize("model,n", [("li_tanh", 256), ("li_tanh", 512), ("li_tanh", 768), ("li_tanh", 1024), ("li_sigmoid", 512),
                                     ("li_tanh", 12), ("li_tanh", 100), ("li_tanh", 500), ("li_tanh", 1000), ("li_sigmoid", 644),
                                     ("li_tanh", 5), ("li_tanh", 50), ("li_tanh", 250), ("li_tanh", 1001), ("li_sigmoid", 125),
                                     ("li_tanh", 1028), ("li_tanh", 1536), ("li_tanh", 1300), ("li_sigmoid", 1280)])
@pytest.mark.parametrize("grad_var", [_abi.RP_OUT_STATE0, _abi.RP_OUT_SOURCE])
def test_resident_adjoint_sweep_equals_a_launch_per_step(hip_device, monkeypatch, model, n, grad_var):
    """The rate models' adjoint (transposed-copy sweep) in ONE launch against one launch per adjoint step: adjoint state, u, input
    gradients and every per-neuron parameter gradient, bit for bit."""
    T = 37
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(T, n, generator=gen) * 0.5).to(hip_device)
    g_out = (torch.randn(T, n, generator=gen) * 0.1).to(hip_device)
    lam_T = (torch.randn(n, generator=gen) * 0.01).to(hip_device)
    got = []
    for resident in ("1", "0"):
        monkeypatch.setenv("RECTIPY_AMD_RESIDENT", resident)
        pop = make(model, n, hip_device)
        pop.transposed_adjoint = True
        tv, tr, ta, ut, gx = (torch.zeros(T, n, device=hip_device) for _ in range(5))
        pop.forward(T, x=x, save_v=tv, save_src=tr, save_acc=ta)
        lam = lam_T.clone()
        gt, gk, ge = (torch.zeros(n, device=hip_device) for _ in range(3))
        ga = {q: torch.zeros(n, device=hip_device) for q in (("r_max", "s", "v0") if model == "li_sigmoid" else ())}
        keep = pop.backward(tv, tr, g_out, grad_var, lam, ut, traj_acc=ta, grad_tau=gt, grad_k=gk, grad_eta=ge, grad_x=gx, grad_params=ga)
        torch.cuda.synchronize()
        del keep
        got.append(dict(lam=lam, u=ut, grad_tau=gt, grad_k=gk, grad_eta=ge, grad_x=gx, **ga))
    assert float(got[0]["lam"].abs().max()) > 0
    for key in got[0]:
        same(got[0][key], got[1][key], f"{model} n={n}: {key}")


@pytest.mark.parametrize("n", [256, 512, 1024, 100, 500, 900, 10, 250, 999, 1028, 1536, 1300])
@pytest.mark.parametrize("grad_var,xw", [(_abi.RP_OUT_SOURCE, "full"), (_abi.RP_OUT_STATE0, "bcast")])
def test_resident_spiking_adjoint_sweep_equals_a_launch_per_step(hip_device, monkeypatch, n, grad_var, xw):
    """The spiking sweep (QIF: resets, surrogate gradient) in ONE launch against one launch per adjoint step, bit for bit."""
    T = 41
    gen = torch.Generator().manual_seed(n + T)
    W = torch.rand(n, n, generator=gen) * 20.0 / n
    params = dict(tau=torch.rand(n, generator=gen) * 0.4 + 0.8, k=torch.rand(n, generator=gen) * 0.4 + 0.8,
                  eta=torch.randn(n, generator=gen) - 5.0, tau_s=torch.rand(n, generator=gen) + 0.5)
    v0 = torch.rand(n, generator=gen) * 65.0 - 5.0
    s0 = torch.rand(n, generator=gen)
    x = (15.0 + torch.randn(T, n if xw == "full" else 1, generator=gen)).to(hip_device)
    g_out = (torch.randn(T, n, generator=gen) * 0.1).to(hip_device)
    lv0, ls0 = torch.randn(n, generator=gen) * 0.01, torch.randn(n, generator=gen) * 0.01
    got = []
    for resident in ("1", "0"):
        monkeypatch.setenv("RECTIPY_AMD_RESIDENT", resident)
        pop = Population("qif", W, params, 1e-3, hip_device, v0=v0, s0=s0, theta=100.0, v_reset=-100.0)
        pop.transposed_adjoint = True
        tv, ts, ta, ut, gx = (torch.zeros(T, n, device=hip_device) for _ in range(5))
        pop.forward(T, x=x, save_v=tv, save_src=ts, save_acc=ta)
        lv, ls = lv0.clone().to(hip_device), ls0.clone().to(hip_device)
        gt, gk, ge, gts = (torch.zeros(n, device=hip_device) for _ in range(4))
        keep = pop.backward(tv, ts, g_out, grad_var, lv, ut, traj_acc=ta, grad_tau=gt, grad_k=gk, grad_eta=ge, grad_x=gx, lam_s=ls,
                            grad_tau_s=gts, slope=0.5, x=x)
        torch.cuda.synchronize()
        del keep
        got.append(dict(lam_v=lv, lam_s=ls, u=ut, grad_tau=gt, grad_k=gk, grad_eta=ge, grad_tau_s=gts, grad_x=gx, v_traj=tv))
    assert int((got[0]["v_traj"] >= 100.0).sum()) > 0   # (spikes inside the window)
    for key in got[0]:
        same(got[0][key], got[1][key], f"qif n={n}: {key}")


@pytest.mark.parametrize("n_in,n_out,T", [(256, 2, 30), (512, 3, 41), (1024, 1, 20), (100, 2, 57), (500, 4, 33), (900, 3, 12), (8, 1, 9),
                                          (5, 1, 12), (50, 2, 30), (250, 3, 25), (301, 4, 19), (511, 2, 14)])   # (n_in % 4 != 0: the 4-byte path)
def test_resident_rls_sweep_equals_the_launches(hip_device, monkeypatch, n_in, n_out, T):
    """RLS with an update every step at launch-bound sizes: the whole sweep in ONE launch (P and W in registers, z exchanged as
    granules) against three launches per update: predictions, losses, weights and P, bit for bit; two consecutive sweeps."""
    from rectipy_amd.rls import RLS
    gen = torch.Generator().manual_seed(n_in + T)
    X = (torch.randn(T, n_in, generator=gen) * 0.3).to(hip_device)
    Y = torch.randn(T, n_out, generator=gen).to(hip_device)
    W0 = torch.randn(n_out, n_in, generator=gen) * 0.1
    got = []
    for resident in ("1", "0"):
        monkeypatch.setenv("RECTIPY_AMD_RESIDENT", resident)
        edge = RLS(n_in, n_out, weights=W0.clone(), beta=0.97, alpha=1.5).to(hip_device)
        half = T // 2
        p1, l1 = edge.sweep(X[:half], Y[:half], 1)
        p2, l2 = edge.sweep(X[half:], Y[half:], 1)
        torch.cuda.synchronize()
        got.append(dict(pred=torch.cat([p1, p2]), loss=torch.cat([l1, l2]), weights=edge.weights.clone(), P=edge.P.clone()))
    for key in got[0]:
        same(got[0][key], got[1][key], f"rls n_in={n_in} n_out={n_out}: {key}")


def test_random_shapes_and_windows_keep_the_bits(hip_device, monkeypatch):
    """Forty random (size, model, window length) draws -- any n up to 1024, both row-dot paths --, three consecutive windows each: the
    resident launch and the launch per step leave the same trajectories and states (a fixed-seed slice of
    scripts/probes/resident_soak.py)."""
    rng = np.random.default_rng(12)
    for case in range(40):
        n = int(rng.integers(1, 257)) * 4 if rng.integers(0, 3) else int(rng.integers(1, 1025))
        model = ["li_tanh", "qif"][int(rng.integers(0, 2))]
        pops = []
        for resident in ("1", "0"):
            monkeypatch.setenv("RECTIPY_AMD_RESIDENT", resident)
            pops.append(make(model, n, hip_device, seed=case))
        for w in range(3):
            T = int(rng.integers(4, 60))
            x = (torch.randn(T, n, generator=torch.Generator().manual_seed(case * 7 + w)) * 0.5 + (1.5 if model == "qif" else 0.0)).to(hip_device)
            outs = []
            for resident, pop in zip(("1", "0"), pops):
                monkeypatch.setenv("RECTIPY_AMD_RESIDENT", resident)
                traj = torch.zeros(T, n, device=hip_device)
                pop.forward(T, x=x, traj=traj, out_var=_abi.RP_OUT_SOURCE)
                outs.append((traj, pop.v.clone(), pop.source.clone()))
            for a, b, name in zip(outs[0], outs[1], ("traj", "v", "source")):
                same(a, b, f"case {case} ({model}, n={n}, window {w}, T={T}): {name}")


@pytest.mark.parametrize("xcd", ["0", "2"], ids=["any-placement", "short-roster"])
def test_the_roster_completes_on_any_placement(hip_device, monkeypatch, xcd):
    """Placement is for speed only: with the one-XCD placement switched off, and with an XCD that "gets" too few workgroups (a test
    switch: only every 32nd workgroup of the chosen XCD takes a ticket, the rest of the roster comes from the other XCDs once every
    workgroup has arrived), the launch gives the bits of one launch per step."""
    n, T = 512, 33
    x = (torch.randn(T, n, generator=torch.Generator().manual_seed(2)) * 0.5).to(hip_device)
    got = []
    for resident in ("1", "0"):
        monkeypatch.setenv("RECTIPY_AMD_RESIDENT", resident)
        monkeypatch.setenv("RECTIPY_AMD_RESIDENT_XCD", xcd)
        pop = make("li_tanh", n, hip_device)
        traj = torch.zeros(T, n, device=hip_device)
        for _ in range(3):
            pop.forward(T, x=x, traj=traj)
        got.append((traj, pop.v.clone()))
    same(got[0][0], got[1][0], "traj")
    same(got[0][1], got[1][1], "v")


def test_without_stream_ordered_scratch_the_launches_take_over(hip_device, monkeypatch):
    """No stream-ordered allocator (a test switch refuses the scratch): nothing is launched resident, the call runs with one launch
    per step -- forward, adjoint sweep and RLS sweep -- and raises nothing."""
    from rectipy_amd.rls import RLS
    from rectipy_amd.engine import default_backend
    n, T = 256, 20
    gen = torch.Generator().manual_seed(4)
    x = (torch.randn(T, n, generator=gen) * 0.5).to(hip_device)
    g_out = (torch.randn(T, n, generator=gen) * 0.1).to(hip_device)
    Y = torch.randn(T, 2, generator=gen).to(hip_device)
    got = []
    for fault in ("2", None):
        if fault:
            monkeypatch.setenv("RECTIPY_AMD_RESIDENT_TEST_FAULT", fault)
        else:
            monkeypatch.delenv("RECTIPY_AMD_RESIDENT_TEST_FAULT", raising=False)
        pop = make("li_tanh", n, hip_device)
        pop.transposed_adjoint = True
        tv, tr, ta, ut = (torch.zeros(T, n, device=hip_device) for _ in range(4))
        pop.forward(T, x=x, save_v=tv, save_src=tr, save_acc=ta)
        lam = torch.zeros(n, device=hip_device)
        keep = pop.backward(tv, tr, g_out, _abi.RP_OUT_SOURCE, lam, ut, traj_acc=ta)
        edge = RLS(n, 2, beta=0.97, alpha=1.5).to(hip_device)
        pred, loss = edge.sweep(tr, Y, 1)
        torch.cuda.synchronize()
        del keep
        got.append(dict(v=pop.v.clone(), lam=lam, u=ut, pred=pred, W=edge.weights.clone()))
    default_backend().async_status()
    for key in got[0]:
        same(got[0][key], got[1][key], key)


def test_a_timed_out_resident_launch_is_reported(hip_device, monkeypatch):
    """A wave that waits in vain for a peer's element gives up after its bounded spin, every wave leaves, and the failure -- the one
    thing a resident launch can only report afterwards -- reaches the host without a synchronisation inside the library: the next
    library call (or rp_async_status after the caller synchronised) raises.  (RECTIPY_AMD_RESIDENT_TEST_FAULT: element 0 is never
    handed on.)"""
    from rectipy_amd.engine import default_backend
    torch.cuda.synchronize()
    default_backend().async_status()   # (nothing pending from the tests before)
    monkeypatch.setenv("RECTIPY_AMD_RESIDENT_TEST_FAULT", "1")
    pop = make("li_tanh", 512, hip_device)
    pop.forward(12, x=torch.zeros(12, 512, device=hip_device))   # returns: the launch is asynchronous
    torch.cuda.synchronize()
    monkeypatch.delenv("RECTIPY_AMD_RESIDENT_TEST_FAULT")
    with pytest.raises(RuntimeError, match="timed out waiting for a peer wave"):
        default_backend().async_status()
    default_backend().async_status()   # (reported once)
    # the library is usable afterwards
    good = make("li_tanh", 512, hip_device)
    good.forward(12, x=torch.zeros(12, 512, device=hip_device))
    torch.cuda.synchronize()
    default_backend().async_status()
    assert bool(torch.isfinite(good.v).all())


def test_no_resident_launch_reported_a_timeout(hip_device):
    """After synchronising: no resident launch of this process (the tests above) timed out waiting for a peer wave."""
    from rectipy_amd.engine import default_backend
    torch.cuda.synchronize()
    default_backend().async_status()


def test_coupled_circuit_takes_the_resident_launch_and_reports_a_peer_that_never_delivers(hip_device, monkeypatch):
    """A circuit of small members runs all steps of an rp_forward_coupled call in one launch (resident_coupled_kernel): with the test
    seam that withholds element 0 of every stream the launch gives up after its bounded wait and the library reports it at the next
    look (rp_async_status) -- so the path was taken --; without the seam the call equals the launches per step (bit for bit:
    tests/test_coupled.py::test_hip_coupled_bit_exact)."""
    from rectipy_amd.engine import default_backend
    from test_coupled import circuit, run_circuit
    sizes, models, out_vars = (64, 96), ("li_tanh", "li_tanh"), (0, 0)
    x = torch.randn(20, sizes[0], generator=torch.Generator().manual_seed(1)).to(hip_device)
    backend = default_backend()
    torch.cuda.synchronize()
    backend.async_status()   # (nothing pending from the tests before)
    monkeypatch.setenv("RECTIPY_AMD_RESIDENT_TEST_FAULT", "1")
    pops, edges = circuit(sizes, models, seed=5, dev=hip_device, backend=None)
    run_circuit(pops, edges, models, out_vars, x, 20, 1, hip_device)
    torch.cuda.synchronize()
    monkeypatch.delenv("RECTIPY_AMD_RESIDENT_TEST_FAULT")
    with pytest.raises(RuntimeError, match="timed out waiting for a peer wave"):
        backend.async_status()
    backend.async_status()   # (reported once)
    pops, edges = circuit(sizes, models, seed=5, dev=hip_device, backend=None)
    res = run_circuit(pops, edges, models, out_vars, x, 20, 1, hip_device)
    torch.cuda.synchronize()
    backend.async_status()
    assert all(bool(torch.isfinite(r[0]).all()) for r in res)


@pytest.mark.parametrize("xcd", ["0", "2"])
def test_coupled_resident_launch_on_any_placement(hip_device, monkeypatch, xcd):
    """The roster of the coupled launch completes wherever its workgroups land: no one-XCD placement (RECTIPY_AMD_RESIDENT_XCD=0)
    and a placement whose XCD "gets" too few workgroups (=2: the others take the leftover tickets) -- the bits of a launch per step."""
    from rectipy_amd.engine import default_backend
    from test_coupled import circuit, run_circuit
    sizes, models, out_vars = (64, 96, 32), ("li_tanh", "qif", "li_tanh"), (0, 1, 1)
    x = (torch.randn(30, sizes[0], generator=torch.Generator().manual_seed(2)) * 0.5).to(hip_device)
    got = []
    for resident in ("1", "0"):
        monkeypatch.setenv("RECTIPY_AMD_RESIDENT", resident)
        monkeypatch.setenv("RECTIPY_AMD_RESIDENT_XCD", xcd)
        pops, edges = circuit(sizes, models, seed=11, dev=hip_device, backend=None)
        got.append(run_circuit(pops, edges, models, out_vars, x, 30, 2, hip_device))
    torch.cuda.synchronize()
    default_backend().async_status()
    for m in range(len(sizes)):
        for a, b, name in zip(got[0][m], got[1][m], ("traj", "samples", "v", "source")):
            same(a, b, f"member {m} {name}")
