"""Populations joined by dense edges, cycles included, as ONE library call per window (rp_forward_coupled,
include/rectipy_hip.h; the reference's per-step graph walk with feedback edges, rectipy/network.py:962-981, 1322-1357).

CPU (oracle backend): the coupled call against an independent numpy float64 restatement of the reference's walk; the front
end's plan against the step-by-step walk of the same graph; the reference-derived fixture G28 runs through this path in
tests/test_topologies.py.  GPU: the HIP library against the oracle, bit for bit, and at the benchmark's shard shape
against a float64 recomputation on the device."""
import numpy as np
import pytest
import torch

import rectipy_amd.engine as engine
from rectipy_amd import FeedbackNetwork, SpikeResetNet, _abi
from rectipy_amd.engine import Population, forward_coupled

TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
QIF = "neuron_model_templates.spiking_neurons.qif.qif"
TARGETS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture
def target(request, oracle_backend):
    if request.param == "oracle":
        return dict(device="cpu", backend=oracle_backend)
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return dict(device="cuda:0", backend=None)


def assert_same(a, b, what):
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    assert a.shape == b.shape, what
    same = (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))
    assert same.all(), f"{what}: {np.count_nonzero(~same)} of {a.size} differ, max abs {np.abs(a - b).max()}"


# ---- engine level -----------------------------------------------------------------------------------------------------
def circuit(sizes, models, seed, dev, backend, dt=1e-2):
    """Populations of the given sizes / models with an edge between EVERY ordered pair (incl. feedback directions)."""
    gen = torch.Generator().manual_seed(seed)
    pops, edges = [], {}
    for n, model in zip(sizes, models):
        if model == "qif":
            W = torch.rand(n, n, generator=gen) * 8.0 / n
            params = dict(tau=1.0, k=1.0, eta=torch.randn(n, generator=gen) + 2.0, tau_s=torch.rand(n, generator=gen) + 0.5)
            pops.append(Population("qif", W, params, dt, dev, v0=torch.randn(n, generator=gen), s0=torch.rand(n, generator=gen) * 0.1,
                                   theta=5.0, v_reset=-5.0, backend=backend))
        else:
            W = torch.randn(n, n, generator=gen) * 1.2 / n ** 0.5
            params = dict(tau=torch.rand(n, generator=gen) * 2 + 0.5, k=torch.rand(n, generator=gen) + 0.5,
                          eta=torch.randn(n, generator=gen) * 0.1)
            pops.append(Population("li_tanh", W, params, dt, dev, v0=torch.randn(n, generator=gen) * 0.5, backend=backend))
    for t_, nt in enumerate(sizes):
        for s_, ns in enumerate(sizes):
            if s_ != t_:
                edges[(s_, t_)] = (torch.randn(nt, ns, generator=gen) * 0.7 / ns ** 0.5).to(dev)
    return pops, edges


def modes_for(models, s_, t_):
    """The mode the front end gives an edge s_ -> t_ when edges towards later members are feed-forward and edges towards
    earlier members feedback edges (FeedbackNetwork._coupled_plan)."""
    if s_ < t_:
        return _abi.RP_EDGE_OUT
    return _abi.RP_EDGE_LAG if models[s_] == "qif" else _abi.RP_EDGE_LIVE


def run_circuit(pops, edges, models, out_vars, x, T, split, dev, S_every=3, cutoff=2):
    """The window in `split` library calls; returns per member (traj, samples, final v, final source)."""
    n_m = len(pops)
    trajs = [torch.zeros(T, p.rows, device=dev) for p in pops]
    samples = [torch.zeros(T, p.rows, device=dev) for p in pops]
    hist = [[torch.zeros(p.rows, device=dev) for _ in range(3)] for p in pops]
    for m, p in enumerate(pops):  # what a LAG edge reads at step 0 (ring slot (0 - 1) % 3): the member's current output
        hist[m][2].copy_(p.v if out_vars[m] == _abi.RP_OUT_STATE0 else p.source)
        p.reset_sampling()
    idx = [0] * n_m
    bounds = np.linspace(0, T, split + 1).astype(int)
    for a, b in zip(bounds[:-1], bounds[1:]):
        members = []
        for m, p in enumerate(pops):
            run = dict(steps=int(b - a), step0=int(a), out_var=out_vars[m], traj=trajs[m][a:b], cutoff=cutoff, sampling_steps=S_every,
                       samples=samples[m], sample_idx=idx[m], hist=hist[m])
            if m == 0:
                run["x"] = x[a:b]
            members.append(dict(pop=p, run=run, edges=[(edges[(s_, m)], s_, modes_for(models, s_, m)) for s_ in range(n_m) if s_ != m]))
        idx = forward_coupled(members)
    return [(trajs[m], samples[m][:idx[m]], pops[m].v, pops[m].source) for m in range(n_m)]


def numpy_walk(pops, edges, models, out_vars, x, T):
    """float64 restatement of FeedbackNetwork's walk for these members (feedback.py: forward): members in order; an edge
    from an earlier member delivers what that member returned this step, an edge from a later rate member its current
    state, an edge from a later spiking member its output of the previous step."""
    f = lambda t: t.detach().cpu().numpy().astype(np.float64)
    st = []
    for p in pops:
        st.append(dict(v=f(p.v), s=f(p.source), W=f(p.W), par={k: f(v) for k, v in p.params.items()}))
    outs = [[] for _ in pops]
    lag = [None] * len(pops)
    for t in range(T):
        ret = {}
        for m, p in enumerate(pops):
            S = st[m]
            E = 0.0
            for s_ in range(len(pops)):
                if s_ == m:
                    continue
                Q = st[s_]
                if s_ < m:
                    vec = ret[s_]
                elif models[s_] == "qif":
                    vec = lag[s_] if lag[s_] is not None else (Q["v"] if out_vars[s_] == _abi.RP_OUT_STATE0 else Q["s"])
                else:
                    vec = Q["v"] if out_vars[s_] == _abi.RP_OUT_STATE0 else np.tanh(Q["v"])
                E = E + f(edges[(s_, m)]) @ vec
            I = (f(x[t]) if m == 0 else 0.0) + E
            par, v = S["par"], S["v"]
            if models[m] == "qif":
                s = S["s"]
                spike = (v - p.theta >= 0).astype(np.float64)
                dv = (v * v + par["eta"] + I) / par["tau"] + par["k"] * (S["W"] @ s)
                ds = -s / par["tau_s"] + spike / p.dt
                ret[m] = v.copy() if out_vars[m] == _abi.RP_OUT_STATE0 else s.copy()
                S["v"] = (v + p.dt * dv) * (1 - spike) + spike * p.v_reset
                S["s"] = s + p.dt * ds
            else:
                r = np.tanh(v)
                ret[m] = v.copy() if out_vars[m] == _abi.RP_OUT_STATE0 else r
                S["v"] = v + p.dt * (-v / par["tau"] + par["k"] * (S["W"] @ r) + I + par["eta"])
            outs[m].append(ret[m])
        for m in range(len(pops)):
            lag[m] = ret[m]
    return [np.asarray(o) for o in outs]


CASES = [((24, 40, 16), ("li_tanh", "li_tanh", "li_tanh"), (0, 0, 1)),
         ((32, 20), ("qif", "li_tanh"), (1, 0)),
         ((17, 30, 12), ("li_tanh", "qif", "qif"), (1, 1, 0)),   # widths off the 16-byte path
         ((256, 520), ("li_tanh", "qif"), (0, 1))]


@pytest.mark.parametrize("sizes,models,out_vars", CASES)
def test_coupled_call_follows_the_walk(oracle_backend, sizes, models, out_vars):
    """Oracle: the coupled call == float64 restatement of the front end's walk (fp32 tolerance), and splitting the window
    into several calls changes nothing (sampling state, history buffers and ping-pong indices carry over)."""
    T = 60
    x = torch.randn(T, sizes[0], generator=torch.Generator().manual_seed(9)) * 0.5
    ref = None
    for split in (1, 3):
        pops, edges = circuit(sizes, models, seed=sum(sizes), dev="cpu", backend=oracle_backend)
        if ref is None:
            ref = numpy_walk(pops, edges, models, out_vars, x, T)
        res = run_circuit(pops, edges, models, out_vars, x, T, split, "cpu")
        for m, (traj, samples, v, s) in enumerate(res):
            assert np.abs(traj.numpy() - ref[m]).max() < 5e-4 * max(1.0, np.abs(ref[m]).max()), f"member {m}"
        if split == 1:
            one = res
        else:
            for m in range(len(pops)):
                for a, b, name in zip(one[m], res[m], ("traj", "samples", "v", "source")):
                    assert_same(a, b, f"split member {m} {name}")


def test_single_steps_with_edges_equal_the_coupled_window(oracle_backend):
    """rp_forward with rp_run.edge_* (one step per call, the sources handed over as plain vectors -- the stepwise walk)
    == rp_forward_coupled for the same graph, bit for bit."""
    sizes, models, out_vars = (24, 40), ("li_tanh", "li_tanh"), (0, 0)
    T = 25
    x = torch.randn(T, sizes[0], generator=torch.Generator().manual_seed(3))
    pops, edges = circuit(sizes, models, seed=5, dev="cpu", backend=oracle_backend)
    win = run_circuit(pops, edges, models, out_vars, x, T, 1, "cpu")
    pops, edges = circuit(sizes, models, seed=5, dev="cpu", backend=oracle_backend)
    traj = [torch.zeros(T, n) for n in sizes]
    for t in range(T):
        v_b = pops[1].v.clone()                       # b's current state: what the feedback edge b -> a reads
        pops[0].forward(1, x=x[t:t + 1], traj=traj[0][t:t + 1], edges=[(edges[(1, 0)], v_b)])
        pops[1].forward(1, traj=traj[1][t:t + 1], edges=[(edges[(0, 1)], traj[0][t].clone())])
    for m in range(2):
        assert_same(win[m][0], traj[m], f"member {m}")
        assert_same(win[m][2], pops[m].v, f"member {m} state")


def test_argument_checks(oracle_backend):
    """The library rejects inconsistent circuits before any launch (no GPU needed: validation precedes every HIP call)."""
    import ctypes as C
    from rectipy_amd import _lib
    lib = _lib.load()
    pops, edges = circuit((8, 16), ("li_tanh", "li_tanh"), seed=1, dev="cpu", backend=oracle_backend)
    arr = (_abi.RpMember * 2)()
    runs = []
    for m, p in enumerate(pops):
        run, keep, _ = p._make_run(steps=2, edges=[(edges[(1 - m, m)], None)])
        runs.append((run, keep))
        arr[m].net, arr[m].run = C.pointer(p._net()), C.pointer(run)
        arr[m].edge_source[0], arr[m].edge_mode[0] = 1 - m, _abi.RP_EDGE_LIVE
    arr[0].edge_mode[0] = _abi.RP_EDGE_OUT           # member 0 cannot read what member 1 returns later in the same step
    assert lib.rp_forward_coupled(arr, 2, None) == _abi.RP_ERR_ARG and b"stepped earlier" in lib.rp_last_error()
    arr[0].edge_mode[0] = _abi.RP_EDGE_LAG            # needs the source's history buffers
    assert lib.rp_forward_coupled(arr, 2, None) == _abi.RP_ERR_ARG and b"hist" in lib.rp_last_error()
    arr[0].edge_mode[0] = _abi.RP_EDGE_LIVE
    arr[0].edge_source[0] = 0                         # its own state while it is stepped
    assert lib.rp_forward_coupled(arr, 2, None) == _abi.RP_ERR_SHAPE  # (and the width no longer matches)
    arr[0].edge_source[0] = 5
    assert lib.rp_forward_coupled(arr, 2, None) == _abi.RP_ERR_ARG
    arr[0].edge_source[0] = 1
    runs[1][0].steps = 3
    assert lib.rp_forward_coupled(arr, 2, None) == _abi.RP_ERR_ARG and b"steps" in lib.rp_last_error()
    assert lib.rp_forward_coupled(None, 2, None) == _abi.RP_ERR_ARG


# ---- HIP vs oracle ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("launches", ["resident", "shared", "member"])
@pytest.mark.parametrize("sizes,models,out_vars", CASES + [((1024, 2052, 512), ("li_tanh", "qif", "li_tanh"), (0, 1, 1)),
                                                           ((4100, 260), ("li_tanh", "li_tanh"), (0, 0)),
                                                           # the resident launch's shapes: two chunks per slot, four slots, ragged widths
                                                           ((256, 512), ("li_tanh", "qif"), (0, 1)),
                                                           ((512, 512), ("li_tanh", "li_tanh"), (1, 1)),
                                                           ((500, 300), ("qif", "li_tanh"), (1, 0)),
                                                           ((128, 132, 64, 32), ("li_tanh", "qif", "li_tanh", "qif"), (0, 0, 1, 1)),
                                                           ((256, 8, 200), ("qif", "qif", "li_tanh"), (0, 1, 0)),
                                                           # ... four rows per wave: slots of three / four chunks, three and four slots of two
                                                           ((1024, 1024), ("li_tanh", "li_tanh"), (0, 1)),
                                                           ((768, 520), ("qif", "li_tanh"), (1, 0)),
                                                           ((1000, 600), ("li_tanh", "qif"), (0, 0)),
                                                           ((512, 260, 384), ("li_tanh", "li_tanh", "qif"), (1, 0, 1)),
                                                           ((300, 512, 256, 128), ("qif", "li_tanh", "li_tanh", "li_tanh"), (0, 1, 0, 1))])
def test_hip_coupled_bit_exact(oracle_backend, hip_device, monkeypatch, sizes, models, out_vars, launches):
    """HIP == oracle, bit for bit, all three ways the library issues the steps: ALL steps of the call in one launch (r05: circuits
    of members of at most 1024 neurons on the 16-byte path, (1 + in-edges) x chunks <= 8; the others fall through to ...) ONE
    launch per step for all members (every edge of these circuits reads a value from before the step; members of at most 4096
    neurons), and one launch per member and step (RECTIPY_AMD_COUPLED_LAUNCHES=member; the 4100-neuron case takes it by size)."""
    if launches == "member":
        monkeypatch.setenv("RECTIPY_AMD_COUPLED_LAUNCHES", "member")
    if launches != "resident":
        monkeypatch.setenv("RECTIPY_AMD_RESIDENT", "0")
    T = 40
    x = torch.randn(T, sizes[0], generator=torch.Generator().manual_seed(9)) * 0.5
    res = []
    for dev, backend in (("cpu", oracle_backend), (hip_device, None)):
        pops, edges = circuit(sizes, models, seed=sum(sizes), dev=dev, backend=backend)
        res.append(run_circuit(pops, edges, models, out_vars, x.to(dev), T, 2, dev))
    for m in range(len(sizes)):
        for a, b, name in zip(res[0][m], res[1][m], ("traj", "samples", "v", "source")):
            assert_same(a, b, f"member {m} {name}")


@pytest.mark.gpu
def test_hip_single_step_edges_bit_exact(oracle_backend, hip_device):
    """rp_forward with edges whose sources are a constant vector and a [T, n] window of vectors."""
    N, n1, n2, T = 520, 260, 77, 12
    gen = torch.Generator().manual_seed(2)
    W = torch.randn(N, N, generator=gen) / N ** 0.5
    params = dict(tau=2.0, k=1.0, eta=0.0)
    E1, E2 = torch.randn(N, n1, generator=gen) / n1 ** 0.5, torch.randn(N, n2, generator=gen) / n2 ** 0.5
    s1, s2 = torch.randn(n1, generator=gen), torch.randn(T, n2, generator=gen)
    x = torch.randn(T, 1, generator=gen)
    res = []
    for dev, backend in (("cpu", oracle_backend), (hip_device, None)):
        pop = Population("li_tanh", W, params, 1e-2, dev, backend=backend)
        traj = torch.zeros(T, N, device=dev)
        pop.forward(T, x=x.to(dev), traj=traj, edges=[(E1.to(dev), s1.to(dev)), (E2.to(dev), s2.to(dev))])
        res.append((traj, pop.v))
    assert_same(res[0][0], res[1][0], "traj")
    assert_same(res[0][1], res[1][1], "v")


@pytest.mark.gpu
def test_full_size_coupled_step_against_float64(hip_device):
    """The benchmark's shape -- two populations of 16384 neurons, a -> b and feedback b -> a, 4 x 1 GiB of matrices per
    Euler step -- for 3 steps against a float64 recomputation on the device."""
    n, T = 16384, 3
    gen = torch.Generator(device=hip_device).manual_seed(7)
    mk = lambda r, c, g: torch.randn(r, c, device=hip_device, generator=gen) * g / c ** 0.5
    W = [mk(n, n, 1.2), mk(n, n, 1.2)]
    E_ab, E_ba = mk(n, n, 0.6), mk(n, n, 0.6)
    x = torch.randn(T, n, device=hip_device, generator=gen) * 0.3
    v0 = [torch.randn(n, device=hip_device, generator=gen) * 0.5 for _ in range(2)]
    pops = [Population("li_tanh", W[m], dict(tau=2.0, k=1.0, eta=0.05), 1e-2, hip_device, v0=v0[m]) for m in range(2)]
    traj = [torch.zeros(T, n, device=hip_device) for _ in range(2)]
    hist = [torch.zeros(n, device=hip_device) for _ in range(3)]
    forward_coupled([
        dict(pop=pops[0], run=dict(steps=T, x=x, traj=traj[0], hist=hist), edges=[(E_ba, 1, _abi.RP_EDGE_LIVE)]),
        dict(pop=pops[1], run=dict(steps=T, traj=traj[1]), edges=[(E_ab, 0, _abi.RP_EDGE_OUT)])])
    v = [t.double() for t in v0]
    for t in range(T):
        va, vb = v
        out_a = va.clone()
        assert (traj[0][t].double() - va).abs().max() < 2e-5 and (traj[1][t].double() - vb).abs().max() < 2e-5, t
        va_new = va + 1e-2 * (-va / 2.0 + W[0].double() @ torch.tanh(va) + (x[t].double() + E_ba.double() @ vb) + 0.05)
        vb_new = vb + 1e-2 * (-vb / 2.0 + W[1].double() @ torch.tanh(vb) + E_ab.double() @ out_a + 0.05)
        v = [va_new, vb_new]
    assert (pops[0].v.double() - v[0]).abs().max() < 2e-5 and (pops[1].v.double() - v[1]).abs().max() < 2e-5


# ---- front end --------------------------------------------------------------------------------------------------------
def build_net(target, force, sizes=(12, 20, 8), spiking=(False, False, False), m=3, k=2, out_r=False, out_act="tanh"):
    net = FeedbackNetwork(1e-2, **target)
    r = np.random.default_rng(1)
    if m:
        net.add_func_node("inp", m, "identity")
    labs = "abc"[:len(sizes)]
    for lab, n, sp in zip(labs, sizes, spiking):
        W = r.standard_normal((n, n)) / np.sqrt(n)
        if sp:
            node = SpikeResetNet.from_pyrates(
                QIF, input_var="qif_op/I_ext", output_var="qif_op/s", weights=W * 5, source_var="qif_op/s",
                target_var="qif_op/s_in", spike_var="qif_op/spike", reset_var="qif_op/v",
                node_vars={"all/qif_op/eta": r.standard_normal(n) + 3, "all/qif_op/v": r.standard_normal(n)},
                spike_threshold=10.0, spike_reset=-10.0, dt=1e-2, dtype=torch.float32, **target)
            net.add_node(lab, node, node_type="diff_eq")
        else:
            net.add_diffeq_node(lab, TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                                output_var="tanh_op/r" if out_r else "li_op/v", dtype=torch.float32,
                                node_vars={"all/li_op/v": r.standard_normal(n) * 0.5, "all/li_op/tau": r.uniform(0.5, 2, n)})
    if k:
        net.add_func_node("out", k, out_act)
    order = (["inp"] if m else []) + list(labs) + (["out"] if k else [])
    width = dict(zip(labs, sizes), inp=m, out=k)
    for s_, t_ in zip(order[:-1], order[1:]):
        net.add_edge(s_, t_, weights=r.standard_normal((width[t_], width[s_])) * 0.5)
    net.add_edge(labs[-1], labs[0], weights=r.standard_normal((sizes[0], sizes[-1])) * 0.5, feedback=True)
    if len(labs) > 2:
        net.add_edge(labs[0], labs[2], weights=r.standard_normal((sizes[2], sizes[0])) * 0.5, feedback=True)
    net._force_stepwise = force
    return net


CONFIGS = [dict(), dict(spiking=(True, False, True)), dict(sizes=(16, 8), spiking=(True, True), m=0, k=0),
           dict(out_r=True, m=0), dict(sizes=(16, 24), spiking=(False, True), out_act="identity"),
           dict(sizes=(16, 24), k=3, out_act="softmax")]


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("cfg", CONFIGS, ids=[str(i) for i in range(len(CONFIGS))])
def test_run_takes_one_call_per_window_and_matches_the_walk(target, cfg, monkeypatch):
    """`FeedbackNetwork.run` on graphs with cycles: ONE coupled library call per window (two with spiking members: the last
    step exposes `y`), same outputs, recordings and final state as the step-by-step walk; state, sampling phase and the
    lagging `y` of spiking members carry from one run to the next."""
    calls = []
    real = engine.forward_coupled
    monkeypatch.setattr(engine, "forward_coupled", lambda members: calls.append(len(members)) or real(members))
    T = 57
    m = cfg.get("m", 3)
    sizes = cfg.get("sizes", (12, 20, 8))
    spiking = cfg.get("spiking", (False,) * len(sizes))
    x = np.random.default_rng(0).standard_normal((T, m if m else sizes[0])).astype(np.float32)
    var = "qif_op/v" if spiking[0] else "li_op/v"
    res = []
    for force in (True, False):
        calls.clear()
        net = build_net(target, force, **cfg)
        o1 = net.run(x[:30], sampling_steps=3, cutoff=4, verbose=False, enable_grad=False, record_vars=[("a", var, False)])
        o2 = net.run(x[30:], sampling_steps=2, verbose=False, enable_grad=False)
        res.append((o1.to_numpy("out"), o2.to_numpy("out"), o1.to_numpy(("a", var)),
                    {k_: v_.detach().cpu().numpy().copy() for k_, v_ in net.state.items()}))
        per_run = 2 if any(spiking) else 1
        assert calls == ([] if force else [len(sizes)] * (2 * per_run)), calls
    for a, b in zip(res[0][:3], res[1][:3]):
        assert a.shape == b.shape and np.abs(a - b).max() < 5e-6
    for k_ in res[0][3]:
        assert np.abs(res[0][3][k_] - res[1][3][k_]).max() < 5e-6, k_


def variant_net(target, force, kind):
    """build_net's two populations a (12) -> b (20) with feedback b -> a, the feed-forward edge a -> b replaced by `kind`."""
    r = np.random.default_rng(5)
    net = build_net(target, force, sizes=(12, 20), k=2)
    net.pop_edge("a", "b")
    W = r.standard_normal((20, 12)) * 0.4
    if kind == "activation":   # a -> tanh node -> b
        net.add_func_node("mid", 16, "tanh")
        net.add_edge("a", "mid", weights=r.standard_normal((16, 12)) * 0.5)
        net.add_edge("mid", "b", weights=r.standard_normal((20, 16)) * 0.4)
    elif kind == "sigmoid_fb":  # ... and a feedback edge from a sigmoid node behind b back into a
        net.add_edge("a", "b", weights=W)
        net.add_func_node("sig", 9, "sigmoid")
        net.add_edge("b", "sig", weights=r.standard_normal((9, 20)) * 0.5)
        net.add_edge("sig", "a", weights=r.standard_normal((12, 9)) * 0.5, feedback=True)
        net.add_edge("sig", "out", weights=r.standard_normal((2, 9)) * 0.5)
    elif kind == "delay":
        net.add_edge("a", "b", weights=W, delays=(np.arange(12) * 5) % 4 + 1)
    elif kind == "filter":
        net.add_edge("a", "b", weights=W, filter_weights=r.standard_normal((12, 12)) * 0.25)
    elif kind == "delay_filter":
        net.add_edge("a", "b", weights=W, delays=np.asarray([1, 3] * 6), filter_weights=r.standard_normal((12, 12)) * 0.25)
    elif kind == "masked":
        net.add_edge("a", "b", weights=W, mask=(r.uniform(size=(20, 12)) < 0.5).astype(np.float32))
    elif kind == "filter_fb":   # the edge with memory on the FEEDBACK edge of the cycle
        net.add_edge("a", "b", weights=W)
        net.pop_edge("b", "a")
        net.add_edge("b", "a", weights=r.standard_normal((12, 20)) * 0.4, filter_weights=r.standard_normal((20, 20)) * 0.2, feedback=True)
    elif kind == "many":        # eleven edges into b: more than RP_MAX_EDGES
        net.add_edge("a", "b", weights=W)
        for q in range(10):
            net.add_diffeq_node(f"p{q}", TANH, weights=r.standard_normal((6, 6)) * 0.4, source_var="tanh_op/r", target_var="li_op/r_in",
                                input_var="li_op/I_ext", output_var="li_op/v", dtype=torch.float32,
                                node_vars={"all/li_op/v": r.standard_normal(6) * 0.5})
            net.add_edge("a", f"p{q}", weights=r.standard_normal((6, 12)) * 0.4)
            net.add_edge(f"p{q}", "b", weights=r.standard_normal((20, 6)) * 0.4)
    else:
        raise ValueError(kind)
    return net


VARIANTS = ["activation", "sigmoid_fb", "delay", "filter", "delay_filter", "masked", "filter_fb", "many"]


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("kind", VARIANTS)
def test_activation_nodes_and_edges_with_memory_stay_inside_the_library(target, kind, monkeypatch, recwarn):
    """Graphs the per-step walk used to take (r04): an activation node BETWEEN populations, a feedback edge from one, an edge
    with memory (delay / filter / both) or a mask between populations -- feed-forward or on the cycle's feedback edge --, more
    than RP_MAX_EDGES edges into one population.  Now ONE coupled library call per window, with the outputs, final states and
    edge memories of the step-by-step walk of the same graph (two runs: state and memory carry over); no warning."""
    calls = []
    real = engine.forward_coupled
    monkeypatch.setattr(engine, "forward_coupled", lambda members: calls.append(len(members)) or real(members))
    x = np.random.default_rng(3).standard_normal((23, 3)).astype(np.float32)
    res = []
    for force in (False, True):
        net = variant_net(target, force, kind)
        n0 = len(calls)
        a = net.run(x[:14], verbose=False, enable_grad=False, sampling_steps=2).to_numpy("out")
        b = net.run(x[14:], verbose=False, enable_grad=False).to_numpy("out")
        assert (len(calls) - n0 == 2) == (not force), (kind, force, calls)
        mems = {}
        for s_, t_ in net.graph.edges:
            edge = net.get_edge(s_, t_)
            for name in ("buffer", "y"):
                if hasattr(edge, name):
                    mems[(s_, t_, name)] = getattr(edge, name).detach().cpu().numpy().copy()
        res.append((a, b, {k_: v.detach().cpu().numpy().copy() for k_, v in net.state.items()}, mems))
    for q in (0, 1):
        assert np.abs(res[0][q] - res[1][q]).max() < 2e-5, kind
    for k_ in res[1][2]:
        assert np.abs(res[0][2][k_] - res[1][2][k_]).max() < 2e-5, k_
    for k_ in res[1][3]:
        assert np.abs(res[0][3][k_] - res[1][3][k_]).max() < 2e-5, k_
    from rectipy_amd import PerformanceWarning
    assert not [w for w in recwarn.list if issubclass(w.category, PerformanceWarning)]


def variant_trained(target, stepwise_adjoint, kind):
    """`variant_net` with every edge trained (input layer, the edges between the members incl. the varied one, the feedback edge,
    the read-out): the gradient has to pass through the populations AND through the varied element."""
    net = variant_net(target, False, kind)
    for s_, t_ in net.graph.edges:
        edge = net.get_edge(s_, t_)
        edge.weights.requires_grad_(True)
        edge.train_params = [edge.weights]
    net._force_stepwise_adjoint = stepwise_adjoint
    return net


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("kind", ["activation", "sigmoid_fb", "delay", "filter", "masked", "filter_fb"])
def test_training_through_activation_nodes_and_edges_with_memory_is_fused(target, kind, monkeypatch, recwarn):
    """(r05) fit_bptt through a cycle that holds an activation node between populations, a feedback edge from one, a delay or
    filter edge (feed-forward or on the feedback edge) or a masked edge: ONE rp_forward_coupled and ONE rp_backward_coupled call
    per window -- the gradient passes through the node (gx = g_out * act'(out)) and through the edge's memory (filter:
    ly(t) = W_e^T gx(t) + F^T ly(t + 1); delay: the total of W_e^T gx(t) into one element d steps earlier) inside the library --
    and the parameters after three SGD windows equal those of autograd over the per-step walk.  No PerformanceWarning."""
    fwd, bwd = [], []
    real_f, real_b = engine.forward_coupled, engine.backward_coupled
    monkeypatch.setattr(engine, "forward_coupled", lambda members: fwd.append(len(members)) or real_f(members))
    monkeypatch.setattr(engine, "backward_coupled", lambda members: bwd.append(len(members)) or real_b(members))
    T, win = 64, 20
    r = np.random.default_rng(1)
    x, y = r.standard_normal((T, 3)).astype(np.float32), r.standard_normal((T, 2)).astype(np.float32) * 0.3
    res = []
    for stepwise in (True, False):
        fwd.clear(), bwd.clear()
        net = variant_trained(target, stepwise, kind)
        obs = net.fit_bptt(x, y, optimizer="sgd", lr=0.05, update_steps=win, sampling_steps=1, verbose=False)
        n_mem = 3 if kind in ("activation", "sigmoid_fb") else 2
        assert (fwd, bwd) == (([], []) if stepwise else ([n_mem] * 4, [n_mem] * 3)), (kind, fwd, bwd)
        mems = {}
        for s_, t_ in net.graph.edges:
            for name in ("buffer", "y"):
                if hasattr(net.get_edge(s_, t_), name):
                    mems[(s_, t_, name)] = getattr(net.get_edge(s_, t_), name).detach().cpu().numpy().copy()
        res.append(([p.detach().cpu().numpy().copy() for p in net.parameters()], obs.to_numpy("loss"), obs.to_numpy("out"), mems))
    assert len(res[0][0]) == len(res[1][0]) >= 4
    for a_, b_ in zip(res[0][0], res[1][0]):
        assert a_.shape == b_.shape and np.abs(a_ - b_).max() <= 3e-5 * max(1.0, np.abs(a_).max()), (kind, np.abs(a_ - b_).max())
    assert np.abs(res[0][1] - res[1][1]).max() < 2e-5 * max(1.0, res[0][1].max())
    assert np.abs(res[0][2] - res[1][2]).max() < 3e-5 * max(1.0, np.abs(res[0][2]).max())
    for k_ in res[0][3]:
        assert np.abs(res[0][3][k_] - res[1][3][k_]).max() < 3e-5, k_
    from rectipy_amd import PerformanceWarning
    assert not [w for w in recwarn.list if issubclass(w.category, PerformanceWarning)]


def test_every_fall_back_to_the_walk_is_loud(oracle_backend):
    """What still runs on the per-step Python walk says so, once per network, naming the cause: a normalising activation node
    between populations, a gradient through a cycle that holds an activation node or an edge with memory, recorded
    variables in fit_bptt.  The `_force_stepwise` test seam stays silent."""
    from rectipy_amd import PerformanceWarning
    target = dict(device="cpu", backend=oracle_backend)
    x = np.random.default_rng(0).standard_normal((10, 3)).astype(np.float32)
    net = build_net(target, False)
    assert net.compile() is None and net._coupled_plan([]) is not None
    # a softmax node between two populations: no in-kernel form
    net = build_net(target, False, sizes=(12, 20), k=0)
    net.pop_edge("a", "b")
    net.add_func_node("mid", 12, "softmax")
    net.add_edge("a", "mid", weights=np.eye(12))
    net.add_edge("mid", "b", weights=np.ones((20, 12)) * 0.1)
    net.compile()
    assert net._coupled_plan([]) is None
    with pytest.warns(PerformanceWarning, match="activation node 'mid' .softmax. lies between populations"):
        assert net.run(x, verbose=False, enable_grad=False).to_numpy("out").shape == (10, 20)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error", PerformanceWarning)   # ... once per network
        net.run(x, verbose=False, enable_grad=False)
    # a trained edge next to a delay + filter edge: the forward plan exists, its adjoint does not (delay and filter edges have one)
    net = variant_net(target, False, "delay_filter")
    net.pop_edge("b", "out")
    net.add_edge("b", "out", weights=np.ones((2, 20)) * 0.1, train="gd")
    net.compile()
    assert net._coupled_plan([]) is not None and net._coupled_grad_plan() is None
    with pytest.warns(PerformanceWarning, match="LinearMemoryFilter between populations"):
        net.fit_bptt(x, np.zeros((10, 2), dtype=np.float32), update_steps=5, verbose=False)
    with pytest.warns(PerformanceWarning, match="LinearMemoryFilter between populations"):
        obs = variant_trained(target, False, "delay_filter").run(x, verbose=False)   # enable_grad defaults to True: the walk, with a graph
    assert obs.stacked("out").requires_grad
    # the test seam asks for the walk: silent
    net = build_net(target, True)
    with warnings.catch_warnings():
        warnings.simplefilter("error", PerformanceWarning)
        net.run(x, verbose=False, enable_grad=False)


@pytest.mark.gpu
def test_matrices_move_into_one_allocation_and_stay_live(hip_device, monkeypatch):
    """The matrices of a coupled plan are consolidated into one device allocation (placement decides 5-8 % of the step
    time at HBM-bound sizes, FeedbackNetwork._consolidate); the tensors nodes and edges hand out stay live views: the
    results do not change, and an in-place edit of `node["weights"]` / `edge.weights` afterwards is seen by the next run."""
    monkeypatch.setattr(FeedbackNetwork, "ARENA_MIN_BYTES", 0)
    target = dict(device="cuda:0", backend=None)
    x = np.random.default_rng(0).standard_normal((20, 16)).astype(np.float32)
    net = build_net(target, False, sizes=(16, 24), m=0, k=0)
    ref = build_net(target, True, sizes=(16, 24), m=0, k=0)
    a = net.run(x, verbose=False, enable_grad=False).to_numpy("out")
    b = ref.run(x, verbose=False, enable_grad=False).to_numpy("out")
    assert np.array_equal(a, b)
    held = [net.get_node("a")["weights"], net.get_node("b")["weights"], net.get_edge("a", "b").weights, net.get_edge("b", "a").weights]
    assert len({t.untyped_storage().data_ptr() for t in held}) == 1 and all(t.data_ptr() % 256 == 0 for t in held)
    arena = net._arena
    with torch.no_grad():
        for n_ in (net, ref):
            n_.get_node("a")["weights"].mul_(0.5)
            n_.get_edge("b", "a").weights.add_(0.1)
    a = net.run(x, verbose=False, enable_grad=False).to_numpy("out")
    b = ref.run(x, verbose=False, enable_grad=False).to_numpy("out")
    assert np.array_equal(a, b) and net._arena is arena  # (nothing to consolidate the second time)


@pytest.mark.gpu
def test_trained_matrices_move_too_and_keep_their_tensor_objects(hip_device, monkeypatch):
    """fit_bptt through a cycle: the trained matrices are consolidated as well -- the parameter objects the optimiser holds
    stay the same, their storage becomes the arena -- and training ends where it ends without consolidation."""
    r = np.random.default_rng(0)
    x, y = r.standard_normal((40, 3)).astype(np.float32), r.standard_normal((40, 2)).astype(np.float32) * 0.3
    target = dict(device="cuda:0", backend=None)
    res = []
    for arena_min in (1 << 60, 0):
        monkeypatch.setattr(FeedbackNetwork, "ARENA_MIN_BYTES", arena_min)
        net = build_trained(target, False, sizes=(16, 24))
        net.compile()
        before = {id(p) for p in net.parameters()}
        net.fit_bptt(x, y, optimizer="sgd", lr=0.05, update_steps=20, sampling_steps=1, verbose=False)
        assert {id(p) for p in net.parameters()} == before
        mats = [net.get_node("a")["weights"], net.get_node("b")["weights"], net.get_edge("a", "b").weights, net.get_edge("b", "a").weights]
        assert (len({t.untyped_storage().data_ptr() for t in mats}) == 1) == (arena_min == 0)
        assert all(t.requires_grad and t.is_leaf for t in mats)
        res.append([p.detach().cpu().numpy().copy() for p in net.parameters()])
    for a, b in zip(*res):
        assert np.array_equal(a, b)


# ---- against the reference itself (golden G36) ---------------------------------------------------------------------------
@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g36_feedback_network_at_kernel_sizes_matches_the_reference(target, monkeypatch):
    """Golden G36 (tests/golden/make_golden_feedback_large.py): the reference's FeedbackNetwork.run with populations of 260
    and 516 rate neurons (a -> b, feedback b -> a) and a QIF pair of 260 + 260 with inhibitory feedback -- the sizes at which
    the coupled step kernel walks full 256-column chunks plus a ragged one in the recurrent AND the edge matrices.  One
    library call per window (two with spiking members).  Tolerances: a few times the reference's own float32-vs-float64
    deviation (1.2e-6 rate, 1.9e-6 QIF, printed by the generator)."""
    import os
    import sys
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    import recipes
    g = dict(np.load(os.path.join(gold, "g36_feedback_large.npz")))
    calls = []
    real = engine.forward_coupled
    monkeypatch.setattr(engine, "forward_coupled", lambda members: calls.append(len(members)) or real(members))
    edge = lambda rows, cols, seed, gain: (recipes.normal((rows, cols), seed) * np.float32(gain / np.sqrt(cols))).astype(np.float32)
    na, nb, T, dt = int(g["na"]), int(g["nb"]), int(g["T"]), float(g["dt"])
    Pa = recipes.li_network(na, seed=3601, g=0.9, defaults=False)
    Pb = recipes.li_network(nb, seed=3602, g=0.9, defaults=False)
    W_ab, W_ba = edge(nb, na, 3603, 0.6), edge(na, nb, 3604, 0.6)
    x = (0.5 * recipes.normal((T, na), seed=3605)).astype(np.float32)
    for name, arr in (("Wa", Pa["W"]), ("Wb", Pb["W"]), ("W_ab", W_ab), ("W_ba", W_ba), ("x", x)):
        assert np.array_equal(recipes.fingerprint(arr), g["fp_" + name]), f"regenerated {name} differs from the generator's"
    net = FeedbackNetwork(dt, **target)
    for lab, P in (("a", Pa), ("b", Pb)):
        net.add_diffeq_node(lab, TANH, weights=P["W"], source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                            output_var="li_op/v", dtype=torch.float32,
                            node_vars={"all/li_op/tau": P["tau"], "all/li_op/k": P["k"], "all/li_op/eta": P["eta"], "all/li_op/v": P["y0"]})
    net.add_edge("a", "b", weights=W_ab)
    net.add_edge("b", "a", weights=W_ba, feedback=True)
    obs = net.run(x, sampling_steps=1, verbose=False, enable_grad=False)
    assert calls == [2]
    assert np.abs(obs.to_numpy("out") - g["rate_out_f64"]).max() < 6e-6
    assert np.abs(net.state["a"].cpu().numpy() - g["rate_state_a_f64"]).max() < 6e-6
    assert np.abs(net.state["b"].cpu().numpy() - g["rate_state_b_f64"]).max() < 6e-6
    # QIF pair
    nq, Tq, dtq = int(g["nq"]), int(g["Tq"]), float(g["dtq"])
    Wq = [(recipes.uniform((nq, nq), 3610 + i) * np.float32(20.0 / nq)).astype(np.float32) for i in range(2)]
    etaq = [(-5.0 + recipes.normal((nq,), 3612)).astype(np.float32), (12.0 + recipes.normal((nq,), 3613)).astype(np.float32)]
    Wq_ab = (recipes.uniform((nq, nq), 3614) * np.float32(10.0 / nq)).astype(np.float32)
    Wq_ba = (-recipes.uniform((nq, nq), 3615) * np.float32(10.0 / nq)).astype(np.float32)
    for name, arr in (("Wq0", Wq[0]), ("Wq1", Wq[1]), ("etaq0", etaq[0]), ("etaq1", etaq[1]), ("Wq_ab", Wq_ab), ("Wq_ba", Wq_ba)):
        assert np.array_equal(recipes.fingerprint(arr), g["fp_" + name]), f"regenerated {name} differs from the generator's"
    assert float(g["qif_peak_f64"]) > 0.5, "the fixture must contain spikes"
    calls.clear()
    net = FeedbackNetwork(dtq, **target)
    for lab, i in (("a", 0), ("b", 1)):
        node = SpikeResetNet.from_pyrates(
            QIF, input_var="qif_op/I_ext", output_var="qif_op/s", weights=Wq[i], source_var="qif_op/s", target_var="qif_op/s_in",
            spike_var="qif_op/spike", reset_var="qif_op/v",
            node_vars={"all/qif_op/eta": etaq[i], "all/qif_op/v": np.full(nq, -2.0), "all/qif_op/s": np.zeros(nq)},
            spike_threshold=100.0, spike_reset=-100.0, dt=dtq, dtype=torch.float32, **target)
        net.add_node(lab, node, node_type="diff_eq")
    net.add_edge("a", "b", weights=Wq_ab)
    net.add_edge("b", "a", weights=Wq_ba, feedback=True)
    obs = net.run(np.full((Tq, nq), 15.0, dtype=np.float32), sampling_steps=1, verbose=False, enable_grad=False)
    assert calls == [2, 2]
    out = obs.to_numpy("out")
    assert np.abs(out[::10] - g["qif_out_f64"]).max() < 5e-5 and np.abs(out[-1] - g["qif_final_f64"]).max() < 5e-5


# ---- adjoint of a coupled window (rp_backward_coupled) -----------------------------------------------------------------
def torch_walk_rate(pops, edges, out_vars, x, T, leaves, post=()):
    """float64 torch restatement of the walk for rate members, differentiable in `leaves` (per member: v0, W, tau, k, eta;
    per edge: its matrix; the input x of member 0).  `post`: edges (s, m) with s < m that are FEEDBACK edges -- they read the
    source's state as it stands, i.e. after its step of the same Euler step."""
    v = [leaves["v0"][m] for m in range(len(pops))]
    outs = [[] for _ in pops]
    for t in range(T):
        ret = {}
        for m, p in enumerate(pops):
            E = 0.0
            for s_ in range(len(pops)):
                if s_ == m:
                    continue
                if s_ < m and (s_, m) not in post:
                    vec = ret[s_]
                else:
                    vec = v[s_] if out_vars[s_] == _abi.RP_OUT_STATE0 else torch.tanh(v[s_])
                E = E + leaves["edges"][(s_, m)] @ vec
            I = (leaves["x"][t] if m == 0 else 0.0) + E
            r = torch.tanh(v[m])
            ret[m] = v[m] if out_vars[m] == _abi.RP_OUT_STATE0 else r
            v[m] = v[m] + p.dt * (-v[m] / leaves["tau"][m] + leaves["k"][m] * (leaves["W"][m] @ r) + I + leaves["eta"][m])
            outs[m].append(ret[m])
    return [torch.stack(o) for o in outs], v


def coupled_adjoint(pops, edges, out_vars, x, T, g_out, g_vend, dev, models=None, post=(), packed_t=None):
    """Forward with the window stored, then rp_backward_coupled; returns the gradients per member / edge."""
    from rectipy_amd.engine import backward_coupled
    n_m = len(pops)
    models = models or ("li_tanh",) * n_m
    f32 = dict(dtype=torch.float32, device=dev)
    saved = [dict(v=torch.zeros(T, p.rows, **f32), r=torch.zeros(T, p.rows, **f32), acc=torch.zeros(T, p.rows, **f32),
                  inp=torch.zeros(T, p.rows, **f32)) for p in pops]
    hist = [[torch.zeros(p.rows, **f32) for _ in range(3)] for p in pops]
    members = []
    for m, p in enumerate(pops):
        hist[m][2].copy_(p.v if out_vars[m] == _abi.RP_OUT_STATE0 else p.source)  # (what a LAG edge reads at step 0)
        run = dict(steps=T, step0=0, out_var=out_vars[m], save_v=saved[m]["v"], save_src=saved[m]["r"], save_acc=saved[m]["acc"],
                   save_in=saved[m]["inp"], hist=hist[m])
        if m == 0:
            run["x"] = x
        members.append(dict(pop=p, run=run, edges=[(edges[(s_, m)], s_, _abi.RP_EDGE_LIVE if (s_, m) in post else modes_for(models, s_, m))
                                                   for s_ in range(n_m) if s_ != m]))
    forward_coupled(members)
    final = [(p.v.clone(), p.source.clone()) for p in pops]   # (what a post-update read of the last step saw)
    res, bw = [], []
    for m, p in enumerate(pops):
        N = p.rows
        R = dict(lam=g_vend[m].to(**f32).clone(), u=torch.zeros(T, N, **f32), g_out=g_out[m].to(**f32).clone(),
                 tau=torch.zeros(N, **f32), k=torch.zeros(N, **f32), eta=torch.zeros(N, **f32), gx=torch.zeros(T, N, **f32))
        kw = dict(traj_v=saved[m]["v"], traj_r=saved[m]["r"], grad_out=R["g_out"], grad_var=out_vars[m], lam=R["lam"],
                  u_traj=R["u"], traj_acc=saved[m]["acc"], grad_x=R["gx"])
        if models[m] == "qif":
            R.update(lam_s=(g_vend[m] * 0.5).to(**f32).clone(), tau_s=torch.zeros(N, **f32))
            kw.update(lam_s=R["lam_s"], slope=0.5, x=saved[m]["inp"],
                      grad_params={k_: R[k_] for k_ in ("tau", "k", "eta", "tau_s")})
        else:
            kw.update(grad_tau=R["tau"], grad_k=R["k"], grad_eta=R["eta"])
        res.append(R)
        bw_edges = members[m]["edges"]
        if packed_t:   # fixed sparse edges: the transposed matrices in packed form
            bw_edges = [edge[:3] + (packed_t.get((edge[1], m)),) for edge in bw_edges]
        bw.append(dict(pop=p, edges=bw_edges, bwd=kw, final_src=final[m][1]))
    keep = backward_coupled(bw)
    del keep
    for m, p in enumerate(pops):
        res[m]["W"] = p.dt * res[m]["u"].T @ saved[m]["r"]
        res[m]["out"] = saved[m]["v"] if out_vars[m] == _abi.RP_OUT_STATE0 else saved[m]["r"]
        res[m]["v_traj"] = saved[m]["v"]
    g_edges = {}
    for (s_, m) in edges:
        out = res[s_]["out"]
        if (s_, m) in post:  # the output one step later; for the last step: the final one
            last = final[s_][0] if out_vars[s_] == _abi.RP_OUT_STATE0 else final[s_][1]
            out = torch.cat([out[1:], last[None, :]])
        g_edges[(s_, m)] = res[m]["gx"].T @ out
    return res, g_edges


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("sizes,out_vars,post", [((24, 40, 16), (0, 0, 1), ()), ((17, 30), (1, 0), ()), ((64,), (1,), ()),
                                                 ((260, 136, 72), (1, 1, 0), ()),
                                                 # feedback edges from members ALREADY stepped: post-update reads
                                                 ((24, 40, 16), (0, 1, 1), ((0, 2),)), ((17, 30), (1, 0), ((0, 1),)),
                                                 ((260, 136, 72), (1, 0, 1), ((0, 1), (1, 2), (0, 2))), ((1028, 24), (1, 0), ((0, 1),))])
def test_coupled_adjoint_against_float64_autograd(target, sizes, out_vars, post):
    dev, T = target["device"], 23
    models = ("li_tanh",) * len(sizes)
    pops, edges = circuit(sizes, models, 5, dev, target["backend"])
    gen = torch.Generator().manual_seed(9)
    x = (torch.randn(T, sizes[0], generator=gen) * 0.5).to(dev)
    g_out = [torch.randn(T, n, generator=gen) for n in sizes]
    g_vend = [torch.randn(n, generator=gen) for n in sizes]
    d = lambda t: t.detach().cpu().double().clone().requires_grad_(True)
    leaves = dict(v0=[d(p.v) for p in pops], W=[d(p.W) for p in pops], tau=[d(p.params["tau"]) for p in pops],
                  k=[d(p.params["k"]) for p in pops], eta=[d(p.params["eta"]) for p in pops],
                  edges={key: d(We) for key, We in edges.items()}, x=d(x))
    outs, v_end = torch_walk_rate(pops, edges, out_vars, x, T, leaves, post)
    loss = sum((o * g.double()).sum() for o, g in zip(outs, g_out)) + sum((v * g.double()).sum() for v, g in zip(v_end, g_vend))
    loss.backward()
    res, g_edges = coupled_adjoint(pops, edges, out_vars, x, T, g_out, g_vend, dev, post=post)

    def close(got, want, what):
        got, want = got.detach().cpu().double(), want.detach()
        scale = float(want.abs().max()) + 1e-12
        assert float((got - want).abs().max()) <= 2e-4 * scale, f"{what}: {float((got - want).abs().max())} of {scale}"

    for m in range(len(sizes)):
        close(res[m]["lam"], leaves["v0"][m].grad, f"member {m} d/dv0")
        close(res[m]["W"], leaves["W"][m].grad, f"member {m} dW")
        for name in ("tau", "k", "eta"):
            close(res[m][name], leaves[name][m].grad, f"member {m} d{name}")
    close(res[0]["gx"], leaves["x"].grad, "d/dx")
    for key in edges:
        close(g_edges[key], leaves["edges"][key].grad, f"edge {key}")


def build_trained(target, stepwise, sizes=(12, 20, 8), sigmoid=False, out_r=False, fb_to_later=False, spiking=None):
    """inp -> a -> b (-> c) -> out with feedback edges from the last population to the first (and from c to b); everything
    trainable: W, tau, k and eta of the populations, the edges between them, the input layer and the read-out."""
    net = FeedbackNetwork(1e-2, **target)
    r = np.random.default_rng(3)
    net.add_func_node("inp", 3, "tanh")
    labs = "abc"[:len(sizes)]
    SIG = "neuron_model_templates.rate_neurons.leaky_integrator.sigmoid"
    for lab, n, sp in zip(labs, sizes, spiking or (False,) * len(sizes)):
        W = r.standard_normal((n, n)) / np.sqrt(n)
        if sp:
            node = SpikeResetNet.from_pyrates(
                QIF, input_var="qif_op/I_ext", output_var="qif_op/s", weights=np.abs(W) * 3, source_var="qif_op/s",
                target_var="qif_op/s_in", spike_var="qif_op/spike", reset_var="qif_op/v",
                train_params=["weights", "qif_op/tau", "qif_op/k", "qif_op/eta", "qif_op/tau_s"],
                node_vars={"all/qif_op/eta": r.standard_normal(n) + 3, "all/qif_op/v": r.uniform(-2, 9, n), "all/qif_op/s": r.uniform(0, 0.5, n)},
                spike_threshold=10.0, spike_reset=-10.0, dt=1e-2, dtype=torch.float32, **target)
            net.add_node(lab, node, node_type="diff_eq")
        elif sigmoid and lab == "b":
            net.add_diffeq_node(lab, SIG, weights=W, source_var="sigmoid_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                                output_var="li_op/v", dtype=torch.float32,
                                node_vars={"all/li_op/v": r.standard_normal(n) * 0.5, "all/li_op/tau": r.uniform(0.5, 2, n)},
                                train_params=["weights", "li_op/tau", "sigmoid_op/s"])
        else:
            net.add_diffeq_node(lab, TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                                output_var="tanh_op/r" if out_r else "li_op/v", dtype=torch.float32,
                                node_vars={"all/li_op/v": r.standard_normal(n) * 0.5, "all/li_op/tau": r.uniform(0.5, 2, n)},
                                train_params=["weights", "li_op/tau", "li_op/k", "li_op/eta"])
    net.add_func_node("out", 2, "identity")
    order = ["inp"] + list(labs) + ["out"]
    width = dict(zip(labs, sizes), inp=3, out=2)
    for s_, t_ in zip(order[:-1], order[1:]):
        net.add_edge(s_, t_, weights=r.standard_normal((width[t_], width[s_])) * 0.5, train="gd")
    net.add_edge(labs[-1], labs[0], weights=r.standard_normal((sizes[0], sizes[-1])) * 0.5, feedback=True, train="gd")
    if len(labs) > 2:
        net.add_edge("c", "b", weights=r.standard_normal((sizes[1], sizes[2])) * 0.5, feedback=True, train="gd")
    if fb_to_later:  # reads a post-update state: no fused adjoint
        net.add_edge("a", labs[-1], weights=r.standard_normal((sizes[-1], sizes[0])) * 0.5, feedback=True)
    net._force_stepwise_adjoint = stepwise
    return net


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("cfg", [dict(), dict(sizes=(16, 24), out_r=True), dict(sigmoid=True), dict(sizes=(260, 136), lr=1e-3),
                                 dict(spiking=(True, False, True), lr=2e-3), dict(sizes=(16, 24), spiking=(True, True), lr=2e-3),
                                 dict(sizes=(20, 12), spiking=(False, True), lr=2e-3)],
                         ids=["three", "two_rates_out", "sigmoid", "wide", "qif_rate_qif", "qif_pair", "rate_qif"])
def test_fit_bptt_through_cycles_is_one_forward_and_one_adjoint_call_per_window(target, cfg, monkeypatch):
    """FeedbackNetwork.fit_bptt: every window is ONE rp_forward_coupled and ONE rp_backward_coupled call; the parameters
    after three SGD windows (and the recorded losses) equal those of autograd over the per-step walk."""
    fwd, bwd = [], []
    real_f, real_b = engine.forward_coupled, engine.backward_coupled
    monkeypatch.setattr(engine, "forward_coupled", lambda members: fwd.append(len(members)) or real_f(members))
    monkeypatch.setattr(engine, "backward_coupled", lambda members: bwd.append(len(members)) or real_b(members))
    T, win = 70, 20
    r = np.random.default_rng(0)
    x, y = r.standard_normal((T, 3)).astype(np.float32), r.standard_normal((T, 2)).astype(np.float32) * 0.3
    res = []
    for stepwise in (True, False):
        fwd.clear(), bwd.clear()
        net = build_trained(target, stepwise, **{k_: v_ for k_, v_ in cfg.items() if k_ != "lr"})
        obs = net.fit_bptt(x, y, optimizer="sgd", lr=cfg.get("lr", 0.05), update_steps=win, sampling_steps=1, verbose=False)
        n_pop = len(cfg.get("sizes", (12, 20, 8)))
        assert (fwd, bwd) == (([], []) if stepwise else ([n_pop] * 4, [n_pop] * 3)), (fwd, bwd)
        res.append(([p.detach().cpu().numpy().copy() for p in net.parameters()], obs.to_numpy("loss"), obs.to_numpy("out"),
                    {k_: v_.detach().cpu().numpy().copy() for k_, v_ in net.state.items()}))
    assert len(res[0][0]) == len(res[1][0]) >= 10
    for a, b in zip(res[0][0], res[1][0]):
        assert a.shape == b.shape and np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max()), np.abs(a - b).max()
    assert np.abs(res[0][1] - res[1][1]).max() < 1e-5 * max(1.0, res[0][1].max()) and np.abs(res[0][2] - res[1][2]).max() < 2e-5 * max(1.0, np.abs(res[0][2]).max())
    for k_ in res[0][3]:
        assert np.abs(res[0][3][k_] - res[1][3][k_]).max() < 2e-5, k_


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("case", ["iku", "ik_biexp"])
def test_population_mean_models_take_the_fused_adjoint(target, case, monkeypatch):
    """(r05) A cycle with a population-mean spiking population (iku_op / ik_biexp_op, ik.yaml:34-68): the coupled adjoint runs
    as ONE rp_backward_coupled call per window -- the member's adjoint means are one more reduction launch per step -- and
    leaves the parameters autograd over the per-step walk leaves (whose single-population form is pinned to the reference by
    G20 / G21).  inp -> a (mean-coupled IK, parameters and start state of the fixture) -> b (tanh) -> out, feedback b -> a."""
    import os
    from test_spiking_bptt import BASE, MEAN_CASES
    c = MEAN_CASES[case]
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", c["gold"] + ".npz")))
    n, op = g["W"].shape[0], c["op"]
    bwd = []
    real_b = engine.backward_coupled
    monkeypatch.setattr(engine, "backward_coupled", lambda members: bwd.append(len(members)) or real_b(members))
    r = np.random.default_rng(8)
    T, win, nb = 90, 30, 20
    xin = np.asarray(g["x"][:T], dtype=np.float32)   # [T, n]: the fixture's input drives population a
    y = r.standard_normal((T, 2)).astype(np.float32) * 0.1
    Wb, Wab, Wba, Wout = (r.standard_normal(sh) * sc for sh, sc in (((nb, nb), 0.2), ((nb, n), 0.3), ((n, nb), 2.0), ((2, nb), 0.3)))
    res = []
    for stepwise in (True, False):
        bwd.clear()
        net = FeedbackNetwork(float(g["dt"]), **target)
        node_vars = {f"all/{op}/{q}": g[q] for q in c["params"]}
        for i, st in enumerate(c["states"]):
            node_vars[f"all/{op}/{st}"] = g["y0"][i * n:(i + 1) * n]
        node = SpikeResetNet.from_pyrates(
            BASE + c["template"], input_var=f"{op}/I_ext", output_var=f"{op}/s", weights=g["W"], source_var=f"{op}/s",
            target_var=f"{op}/s_in", spike_var=f"{op}/spike", reset_var=f"{op}/v", node_vars=node_vars,
            spike_threshold=float(g["theta"]), spike_reset=float(g["v_reset"]), dt=float(g["dt"]), dtype=torch.float32,
            train_params=["weights"] + [f"{op}/{q}" for q in c["params"]], **target)
        node.enable_spike_recording(8192)
        net.add_node("a", node, node_type="diff_eq")
        net.add_diffeq_node("b", TANH, weights=Wb, source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                            output_var="tanh_op/r", dtype=torch.float32, train_params=["weights", "li_op/tau"])
        net.add_func_node("out", 2, "identity")
        net.add_edge("a", "b", weights=Wab, train="gd")
        net.add_edge("b", "out", weights=Wout, train="gd")
        net.add_edge("b", "a", weights=Wba, feedback=True, train="gd")
        net._force_stepwise_adjoint = stepwise
        obs = net.fit_bptt(xin, y, optimizer="sgd", lr=float(g["lr"]) * 0.1, update_steps=win, sampling_steps=1, verbose=False)
        assert bwd == ([] if stepwise else [2] * 3), bwd
        assert len(node.spikes()) >= 10, "the windows must contain spikes"
        res.append(([p.detach().cpu().numpy().copy() for p in net.parameters()], obs.to_numpy("out")))
    assert len(res[0][0]) == len(res[1][0]) >= 12
    for a_, b_ in zip(res[0][0], res[1][0]):
        assert a_.shape == b_.shape and np.abs(a_ - b_).max() <= 5e-5 * max(1.0, np.abs(a_).max()), np.abs(a_ - b_).max()
    assert np.abs(res[0][1] - res[1][1]).max() < 5e-5 * max(1.0, np.abs(res[0][1]).max())


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("out_r", [False, True])
def test_feedback_from_a_population_already_stepped_takes_the_fused_adjoint_too(target, out_r, monkeypatch):
    """A feedback edge from an EARLIER population delivers its post-update state (network.py:1354-1357): in the adjoint its
    gradient lands one row later in the source's output gradient (for the last step: in the final adjoint).  Same parameters
    after three SGD windows as autograd over the per-step walk."""
    bwd = []
    real_b = engine.backward_coupled
    monkeypatch.setattr(engine, "backward_coupled", lambda members: bwd.append(len(members)) or real_b(members))
    r = np.random.default_rng(0)
    x, y = r.standard_normal((65, 3)).astype(np.float32), r.standard_normal((65, 2)).astype(np.float32) * 0.3
    res = []
    for stepwise in (True, False):
        bwd.clear()
        net = build_trained(target, stepwise, out_r=out_r, fb_to_later=True)
        for s_, t_ in (("a", "c"),):
            net.get_edge(s_, t_).weights.requires_grad_(True)   # (the post-update edge is trained as well)
        assert (net._coupled_grad_plan() is None) == stepwise if net._order else True
        obs = net.fit_bptt(x, y, optimizer="sgd", lr=0.05, update_steps=20, sampling_steps=1, verbose=False)
        assert bwd == ([] if stepwise else [3] * 3), bwd
        res.append(([p.detach().cpu().numpy().copy() for p in net.parameters()], obs.to_numpy("out")))
    for a_, b_ in zip(res[0][0], res[1][0]):
        assert a_.shape == b_.shape and np.abs(a_ - b_).max() <= 2e-5 * max(1.0, np.abs(a_).max()), np.abs(a_ - b_).max()
    assert np.abs(res[0][1] - res[1][1]).max() < 2e-5 * max(1.0, np.abs(res[0][1]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("post", [(), ((0, 1),)], ids=["", "post-update-read"])
@pytest.mark.parametrize("sizes,out_vars,models", [
    ((24, 40, 16), (0, 0, 1), None), ((17, 30), (1, 0), None), ((260, 136, 72), (1, 1, 0), None),
    ((1024, 2052, 512), (0, 1, 1), None), ((4100, 520), (1, 0), None),
    ((32, 20), (1, 0), ("qif", "li_tanh")), ((17, 30, 12), (1, 1, 0), ("li_tanh", "qif", "qif")),
    ((520, 1028), (1, 1), ("qif", "qif")),
    # strips of 256 columns that do not fill, four members
    ((256, 512, 300), (0, 1, 0), None), ((64, 128, 36, 260), (0, 1, 1, 0), ("li_tanh", "qif", "li_tanh", "qif"))])
def test_hip_coupled_adjoint_bit_exact(oracle_backend, hip_device, sizes, out_vars, models, post):
    """rp_backward_coupled on the GPU against the C oracle: adjoint states, per-step u, input gradients, the output
    gradients the edges accumulate into and the per-neuron parameter gradients, bit for bit."""
    T = 19
    models = models or ("li_tanh",) * len(sizes)
    if post and models[0] == "qif":
        pytest.skip("a post-update read needs a rate population as its source")
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(T, sizes[0], generator=gen) * 0.5
    g_out = [torch.randn(T, n, generator=gen) for n in sizes]
    g_vend = [torch.randn(n, generator=gen) for n in sizes]
    got = []
    for dev, backend in (("cpu", oracle_backend), (hip_device, None)):
        pops, edges = circuit(sizes, models, 5, dev, backend, dt=0.08 if "qif" in models else 1e-2)
        res, _ = coupled_adjoint(pops, edges, out_vars, x.to(dev), T, g_out, g_vend, dev, models, post=post)
        got.append(res)
    for m in range(len(sizes)):
        if models[m] == "qif":  # spikes inside the window: the sweep passes resets and surrogate gradients
            assert int((got[0][m]["v_traj"] >= pops[m].theta).sum()) >= 3
        for key in ("lam", "u", "gx", "g_out", "tau", "k", "eta") + (("lam_s", "tau_s") if models[m] == "qif" else ()):
            assert_same(got[0][m][key], got[1][m][key], f"member {m} {key}")


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_run_with_gradient_and_epoch_fit_take_the_fused_adjoint(target, monkeypatch):
    """`run(enable_grad=True, truncate_steps=...)` and `fit_bptt` with list inputs (one update per epoch on the sampled
    outputs): windows of the coupled autograd node; gradients and updates equal those of the per-step walk."""
    bwd = []
    real_b = engine.backward_coupled
    monkeypatch.setattr(engine, "backward_coupled", lambda members: bwd.append(len(members)) or real_b(members))
    r = np.random.default_rng(4)
    T = 45
    x, y = r.standard_normal((T, 3)).astype(np.float32), r.standard_normal((T, 2)).astype(np.float32) * 0.3
    res = []
    for stepwise in (True, False):
        bwd.clear()
        net = build_trained(target, stepwise, sizes=(16, 24))
        obs = net.run(x, sampling_steps=2, cutoff=3, truncate_steps=20, verbose=False, enable_grad=True)
        out = obs.stacked("out")
        (out ** 2).sum().backward()
        grads = [p.grad.detach().cpu().numpy().copy() for p in net.parameters()]
        assert bwd == ([] if stepwise else [2] * 3), bwd   # three windows, cut from each other by truncate_steps
        net = build_trained(target, stepwise, sizes=(16, 24))
        obs2 = net.fit_bptt([x, x[:30]], [y[::2], y[:30:2]], optimizer="sgd", lr=0.05, sampling_steps=2, verbose=False)
        res.append((out.detach().cpu().numpy(), grads, [p.detach().cpu().numpy().copy() for p in net.parameters()],
                    np.asarray(obs2["epoch_loss"], dtype=np.float64)))
    assert np.abs(res[0][0] - res[1][0]).max() < 2e-5
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert a.shape == b.shape and np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max()), np.abs(a - b).max()
    assert np.abs(res[0][3] - res[1][3]).max() < 1e-5


# ---- training through cycles against the reference itself (golden G37) ------------------------------------------------------
def g37():
    import os
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g37_feedback_fit.npz")))


def assert_grad(param, ref, what, rel):
    """The gradient the last window left on the parameter against the reference's, relative to its largest entry."""
    got = param.grad.detach().cpu().numpy().astype(np.float64)
    assert np.abs(got - ref).max() <= rel * np.abs(ref).max(), f"grad {what}: off by {np.abs(got - ref).max() / np.abs(ref).max():.2e} of max"


def assert_update(got, ref, start, what, rel=2e-4):
    """The parameter after training against the reference's: the deviation is measured against the size of the UPDATE."""
    got = got.detach().cpu().numpy().astype(np.float64)
    step = max(np.abs(ref - start).max(), 1e-12)
    assert np.abs(got - ref).max() <= rel * step + 2e-7 * max(1.0, np.abs(ref).max()), \
        f"{what}: off by {np.abs(got - ref).max():.3e}, update {step:.3e}"


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("adjoint", ["fused", "walk"])
def test_g37_rate_cycle_fit_matches_the_reference(target, adjoint, monkeypatch):
    """Golden G37 (a): the reference's FeedbackNetwork.fit_bptt, two SGD windows through a -> b -> out with feedback b -> a,
    every parameter trained.  Both the fused adjoint (one rp_backward_coupled call per window) and the per-step autograd end
    at the reference's float64 parameters (tolerance: 2e-4 of each parameter's update; the reference's own float32 run
    deviates by 1.7e-6 absolute)."""
    g = g37()
    bwd = []
    real_b = engine.backward_coupled
    monkeypatch.setattr(engine, "backward_coupled", lambda members: bwd.append(len(members)) or real_b(members))
    net = FeedbackNetwork(float(g["dt"]), **target)
    names = ["weights", "li_op/tau", "li_op/k", "li_op/eta"]
    for lab in "ab":
        net.add_diffeq_node(lab, TANH, weights=g[f"rate_{lab}0_W"], source_var="tanh_op/r", target_var="li_op/r_in",
                            input_var="li_op/I_ext", output_var="li_op/v", dtype=torch.float32, train_params=list(names),
                            node_vars={"all/li_op/tau": g[f"rate_{lab}0_tau"], "all/li_op/k": g[f"rate_{lab}0_k"],
                                       "all/li_op/eta": g[f"rate_{lab}0_eta"], "all/li_op/v": g[f"rate_{lab}0_y0"]})
    net.add_func_node("out", 3, "identity")
    e_ab = net.add_edge("a", "b", weights=g["rate_W_ab0"], train="gd")
    e_out = net.add_edge("b", "out", weights=g["rate_W_out0"], train="gd")
    e_ba = net.add_edge("b", "a", weights=g["rate_W_ba0"], feedback=True, train="gd")
    net._force_stepwise_adjoint = adjoint == "walk"
    obs = net.fit_bptt(g["rate_x"].astype(np.float32), g["rate_target"].astype(np.float32), optimizer="sgd", lr=float(g["lr"]),
                       update_steps=int(g["win"]), sampling_steps=1, verbose=False)
    assert bwd == ([2, 2] if adjoint == "fused" else [])
    for lab in "ab":
        node = net.get_node(lab)
        for key, start in zip(names, ("W", "tau", "k", "eta")):
            assert_update(node[key], g[f"rate_{lab}_{key.split('/')[-1]}_f64"], g[f"rate_{lab}0_{start}"], f"{lab} {key}")
            assert_grad(node[key], g[f"rate_grad_{lab}_{key.split('/')[-1]}_f64"], f"{lab} {key}", 1e-4)
    for edge, key in ((e_ab, "W_ab"), (e_ba, "W_ba"), (e_out, "W_out")):
        assert_update(edge.weights, g[f"rate_{key}_f64"], g[f"rate_{key}0"], key)
        assert_grad(edge.weights, g[f"rate_grad_{key}_f64"], key, 1e-4)
    assert np.abs(obs.to_numpy("out") - g["rate_pred_f64"]).max() < 1e-5
    assert np.abs(obs.to_numpy("loss") - g["rate_loss_f64"]).max() < 1e-5


def build_g37_qif(g, target):
    net = FeedbackNetwork(float(g["dtq"]), **target)
    names = ["weights", "qif_op/tau", "qif_op/k", "qif_op/eta", "qif_op/tau_s"]
    nq = g["qif_a0_eta"].shape[0]
    for lab in "ab":
        node = SpikeResetNet.from_pyrates(
            QIF, input_var="qif_op/I_ext", output_var="qif_op/s", weights=g[f"qif_{lab}0_W"], source_var="qif_op/s",
            target_var="qif_op/s_in", spike_var="qif_op/spike", reset_var="qif_op/v", train_params=list(names),
            node_vars={"all/qif_op/eta": g[f"qif_{lab}0_eta"], "all/qif_op/tau": g[f"qif_{lab}0_tau"], "all/qif_op/k": g[f"qif_{lab}0_k"],
                       "all/qif_op/tau_s": g[f"qif_{lab}0_tau_s"], "all/qif_op/v": g[f"qif_{lab}0_y0"][:nq],
                       "all/qif_op/s": g[f"qif_{lab}0_y0"][nq:]},
            spike_threshold=float(g["theta"]), spike_reset=float(g["v_reset"]), dt=float(g["dtq"]), dtype=torch.float32, **target)
        net.add_node(lab, node, node_type="diff_eq")
    net.add_func_node("out", 2, "identity")
    edges = dict(W_ab=net.add_edge("a", "b", weights=g["qif_W_ab0"], train="gd"),
                 W_out=net.add_edge("b", "out", weights=g["qif_W_out0"], train="gd"),
                 W_ba=net.add_edge("b", "a", weights=g["qif_W_ba0"], feedback=True, train="gd"))
    return net, names, edges


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("adjoint", ["fused", "walk"])
def test_g37_qif_cycle_fit_matches_the_reference(target, adjoint, monkeypatch):
    """Golden G37 (b): the reference's FeedbackNetwork.fit_bptt through a QIF pair a -> b -> out with feedback b -> a (the edge
    reads b's state of the previous step and the surrogate gradient passes back through that read), one SGD window with
    spikes in both populations, every parameter trained.  Tolerance: 1e-3 of each parameter's update (the reference's own
    float32 run: 3.5e-6 absolute on updates of up to 0.28)."""
    g = g37()
    bwd = []
    real_b = engine.backward_coupled
    monkeypatch.setattr(engine, "backward_coupled", lambda members: bwd.append(len(members)) or real_b(members))
    net, names, edges = build_g37_qif(g, target)
    net._force_stepwise_adjoint = adjoint == "walk"
    obs = net.fit_bptt(g["qif_x"].astype(np.float32), g["qif_target"].astype(np.float32), optimizer="sgd", lr=float(g["lrq"]),
                       update_steps=int(g["winq"]), sampling_steps=1, verbose=False)
    assert bwd == ([2] if adjoint == "fused" else [])
    assert np.abs(obs.to_numpy("out") - g["qif_pred_f64"]).max() < 2e-5
    for lab in "ab":
        node = net.get_node(lab)
        for key, start in zip(names, ("W", "tau", "k", "eta", "tau_s")):
            assert_update(node[key], g[f"qif_{lab}_{key.split('/')[-1]}_f64"], g[f"qif_{lab}0_{start}"], f"{lab} {key}", rel=1e-3)
            assert_grad(node[key], g[f"qif_grad_{lab}_{key.split('/')[-1]}_f64"], f"{lab} {key}", 1e-3)
    for key, edge in edges.items():
        assert_update(edge.weights, g[f"qif_{key}_f64"], g[f"qif_{key}0"], key, rel=1e-3)
        assert_grad(edge.weights, g[f"qif_grad_{key}_f64"], key, 1e-3)


def test_repeated_parameters_are_stepped_one_after_the_other(oracle_backend):
    """FeedbackNetwork.parameters() repeats the nodes at a feedback edge (the reference's list, network.py:1308-1322), so an
    optimiser steps them twice per update.  torch's multi-tensor path (its default on a GPU) would do both steps in one
    kernel, racing on the tensor (seen on MI355X: run-to-run different parameters); the optimiser is built without it."""
    net = build_trained(dict(device="cpu", backend=oracle_backend), False, sizes=(16, 24))
    net.compile()
    params = list(net.parameters())
    assert len(params) == 20 and len({id(p) for p in params}) == 12   # a and b: W, tau, k, eta twice; four edges once
    opt = net._get_optimizer("sgd", 0.1, net.parameters())
    assert opt.defaults["foreach"] is False
    opt = net._get_optimizer("adam", 0.1, net.parameters(), {"foreach": True})   # (an explicit choice stands)
    assert opt.defaults["foreach"] is True
    plain = net._get_optimizer("sgd", 0.1, iter(params[:4]))
    assert plain.defaults["foreach"] is None


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("stride", [1, 3])
def test_fit_rls_behind_coupled_populations_is_windows_plus_one_sweep(target, stride, monkeypatch):
    """fit_rls on a read-out of populations joined in a cycle: the read-out does not feed back, so the populations integrate
    the whole input with one rp_forward_coupled call per window and the RLS sweep runs over the stored edge inputs; same
    predictions, losses, weights, P and final state as the step-by-step walk with one RLS.update per step."""
    calls = []
    real = engine.forward_coupled
    monkeypatch.setattr(engine, "forward_coupled", lambda members: calls.append(len(members)) or real(members))
    r = np.random.default_rng(2)
    x, y = r.standard_normal((41, 3)).astype(np.float32), r.standard_normal((41, 2)).astype(np.float32)
    res = []
    for force in (True, False):
        calls.clear()
        net = build_net(target, force, sizes=(16, 24), k=0)
        net.add_func_node("out", 2, "identity")
        net.add_edge("b", "out", train="rls", beta=1.0, alpha=1.5)
        obs = net.fit_rls(x, y, update_steps=stride, sampling_steps=1, verbose=False)
        edge = net.get_edge("b", "out")
        assert calls == ([] if force else [2]), calls
        res.append((obs.to_numpy("out"), obs.to_numpy("loss"), edge.weights.cpu().numpy().copy(), edge.P.cpu().numpy().copy(),
                    {k_: v_.detach().cpu().numpy().copy() for k_, v_ in net.state.items()}))
    for a, b in zip(res[0][:4], res[1][:4]):
        assert a.shape == b.shape and np.abs(a - b).max() < 2e-5 * max(1.0, np.abs(a).max())
    for k_ in res[0][4]:
        assert np.abs(res[0][4][k_] - res[1][4][k_]).max() < 5e-6, k_


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g38_cycle_fit_at_kernel_sizes_matches_the_reference(target, monkeypatch):
    """Golden G38 (tests/golden/make_golden_feedback_fit_large.py): the reference's FeedbackNetwork.fit_bptt with populations
    of 260 and 516 neurons (a -> b -> read-out, feedback b -> a), one SGD window of 40 steps, every parameter trained -- the
    sizes at which the coupled adjoint's chunked products walk full 256-column strips plus a ragged one in the recurrent and
    the edge matrices.  The gradients the window leaves on the parameters against the reference's float64 autograd: the
    per-neuron vectors whole, eight rows and eight columns of every matrix and its sum / sum of squares.  Tolerance 2e-5 of
    each gradient's largest entry (the reference's own float32 run: 4.9e-7)."""
    import os
    import sys
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    import recipes
    g = dict(np.load(os.path.join(gold, "g38_feedback_fit_large.npz")))
    bwd = []
    real_b = engine.backward_coupled
    monkeypatch.setattr(engine, "backward_coupled", lambda members: bwd.append(len(members)) or real_b(members))
    edge = lambda rows, cols, seed, gain: (recipes.normal((rows, cols), seed) * np.float32(gain / np.sqrt(cols))).astype(np.float32)
    na, nb, T = int(g["na"]), int(g["nb"]), int(g["T"])
    Pa = recipes.li_network(na, seed=3801, g=1.1, defaults=False)
    Pb = recipes.li_network(nb, seed=3802, g=1.1, defaults=False)
    W_ab, W_ba, W_out = edge(nb, na, 3803, 0.6), edge(na, nb, 3804, 0.6), edge(3, nb, 3805, 1.0)
    x = (0.5 * recipes.normal((T, na), seed=3806)).astype(np.float32)
    tgt = (0.2 * recipes.normal((T, 3), seed=3807)).astype(np.float32)
    for name, arr in (("Wa", Pa["W"]), ("Wb", Pb["W"]), ("W_ab", W_ab), ("W_ba", W_ba), ("W_out", W_out), ("x", x), ("target", tgt)):
        assert np.array_equal(recipes.fingerprint(arr), g["fp_" + name]), f"regenerated {name} differs from the generator's"
    net = FeedbackNetwork(float(g["dt"]), **target)
    names = ["weights", "li_op/tau", "li_op/k", "li_op/eta"]
    for lab, P in (("a", Pa), ("b", Pb)):
        net.add_diffeq_node(lab, TANH, weights=P["W"], source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                            output_var="li_op/v", dtype=torch.float32, train_params=list(names),
                            node_vars={"all/li_op/tau": P["tau"], "all/li_op/k": P["k"], "all/li_op/eta": P["eta"], "all/li_op/v": P["y0"]})
    net.add_func_node("out", 3, "identity")
    e_ab = net.add_edge("a", "b", weights=W_ab, train="gd")
    e_out = net.add_edge("b", "out", weights=W_out, train="gd")
    e_ba = net.add_edge("b", "a", weights=W_ba, feedback=True, train="gd")
    obs = net.fit_bptt(x, tgt, optimizer="sgd", lr=float(g["lr"]), update_steps=T, sampling_steps=1, verbose=False)
    assert bwd == [2]
    assert np.abs(obs.to_numpy("out") - g["pred_f64"]).max() < 1e-5
    rows = {"a": g["rows_a"], "b": g["rows_b"]}
    for lab in "ab":
        for key in names[1:]:
            assert_grad(net.get_node(lab)[key], g[f"grad_{lab}_{key.split('/')[-1]}_f64"], f"{lab} {key}", 2e-5)
    mats = {"Wa": (net.get_node("a")["weights"], "a", "a"), "Wb": (net.get_node("b")["weights"], "b", "b"), "W_ab": (e_ab.weights, "b", "a"),
            "W_ba": (e_ba.weights, "a", "b"), "W_out": (e_out.weights, None, "b")}
    for key, (param, rlab, clab) in mats.items():
        got = param.grad.detach().cpu().numpy().astype(np.float64)
        ref_rows, ref_cols, fp_ref = g[f"grad_{key}_rows_f64"], g[f"grad_{key}_cols_f64"], g[f"grad_{key}_fp_f64"]
        scale = max(np.abs(ref_rows).max(), np.abs(ref_cols).max())
        assert np.abs((got[rows[rlab]] if rlab else got) - ref_rows).max() <= 2e-5 * scale, key
        assert np.abs(got[:, rows[clab]] - ref_cols).max() <= 2e-5 * scale, key
        fp = recipes.fingerprint(got)
        assert abs(fp[1] - fp_ref[1]) <= 1e-5 * fp_ref[1], key   # sum of squares of all entries


# ---- packed sparse matrices inside coupled populations (rp_run.edge_sp) -----------------------------------------------------
def sparse_circuit(sizes, models, seed, dev, backend, density=0.1, dt=1e-2):
    """`circuit` with every matrix thinned to `density` (zeros elsewhere), as utility.random_connectivity leaves them."""
    gen = torch.Generator().manual_seed(seed)
    pops, edges = [], {}
    for n, model in zip(sizes, models):
        mask = (torch.rand(n, n, generator=gen) < density).float()
        if model == "qif":
            W = torch.rand(n, n, generator=gen) * 8.0 / (n * density) * mask
            params = dict(tau=1.0, k=1.0, eta=torch.randn(n, generator=gen) + 2.0, tau_s=torch.rand(n, generator=gen) + 0.5)
            pops.append(Population("qif", W, params, dt, dev, v0=torch.randn(n, generator=gen), s0=torch.rand(n, generator=gen) * 0.1,
                                   theta=5.0, v_reset=-5.0, backend=backend))
        else:
            W = torch.randn(n, n, generator=gen) * 1.2 / (n * density) ** 0.5 * mask
            params = dict(tau=torch.rand(n, generator=gen) * 2 + 0.5, k=torch.rand(n, generator=gen) + 0.5,
                          eta=torch.randn(n, generator=gen) * 0.1)
            pops.append(Population("li_tanh", W, params, dt, dev, v0=torch.randn(n, generator=gen) * 0.5, backend=backend))
    for t_, nt in enumerate(sizes):
        for s_, ns in enumerate(sizes):
            if s_ != t_:
                mask = (torch.rand(nt, ns, generator=gen) < density).float()
                edges[(s_, t_)] = (torch.randn(nt, ns, generator=gen) * 0.7 / (ns * density) ** 0.5 * mask).to(dev)
    return pops, edges


@pytest.mark.gpu
@pytest.mark.parametrize("pack_w", [True, False], ids=["packed-W", "dense-W"])
@pytest.mark.parametrize("sizes,models,out_vars", [((1024, 2052), ("li_tanh", "li_tanh"), (0, 1)), ((520, 260, 1028), ("li_tanh", "qif", "li_tanh"), (1, 1, 0)),
                                                   ((4096, 4100), ("qif", "li_tanh"), (1, 0))])
def test_hip_packed_edges_in_coupled_members_bit_exact(oracle_backend, hip_device, sizes, models, out_vars, pack_w):
    """Members whose W and edge matrices are all in packed sparse form (rp_run.edge_sp): one packed mat-vec launch per edge,
    then the member's packed step -- against the C oracle's DENSE arithmetic on the same (mostly zero) matrices, bit for
    bit: trajectories, samples, final states; per-row and blocked packed forms, ragged chunks, three edges per member."""
    from rectipy_amd import sparse
    T = 23
    x = torch.randn(T, sizes[0], generator=torch.Generator().manual_seed(3)) * 0.5 + (1.0 if models[0] == "qif" else 0.0)
    res = []
    for dev, backend in (("cpu", oracle_backend), (hip_device, None)):
        pops, edges = sparse_circuit(sizes, models, 11, dev, backend)
        packed = {}
        if dev != "cpu":
            for p in pops:
                if pack_w:   # (else: dense recurrent matrices, sparse coupling between the populations)
                    p.pack_weights()
                assert (p.packed is not None) == pack_w
            for (s_, t_), We in edges.items():
                rb = pops[t_]._block_rows()
                packed[(s_, t_)] = sparse.pack_rows_blocked(We, block_rows=rb) if rb > 0 else sparse.pack_rows(We)
        n_m = len(pops)
        trajs = [torch.zeros(T, p.rows, device=dev) for p in pops]
        hist = [[torch.zeros(p.rows, device=dev) for _ in range(3)] for p in pops]
        for m, p in enumerate(pops):
            hist[m][2].copy_(p.v if out_vars[m] == _abi.RP_OUT_STATE0 else p.source)
        for a, b in ((0, 9), (9, T)):
            members = []
            for m, p in enumerate(pops):
                run = dict(steps=b - a, step0=a, out_var=out_vars[m], traj=trajs[m][a:b], hist=hist[m])
                if m == 0:
                    run["x"] = x[a:b].to(dev)
                members.append(dict(pop=p, run=run, edges=[(edges[(s_, m)], s_, modes_for(models, s_, m), packed.get((s_, m)))
                                                           for s_ in range(n_m) if s_ != m]))
            forward_coupled(members)
        if dev != "cpu":  # (the packed path leaves the node input of the last step in the members' edge_sum workspace)
            assert all(float(p._edge_sum.abs().sum()) > 0 for p in pops)
        res.append([(trajs[m], pops[m].v, pops[m].source) for m in range(n_m)])
    for m in range(len(sizes)):
        for a, b, what in zip(res[0][m], res[1][m], ("trajectory", "v", "source")):
            assert_same(a, b, f"member {m} {what}")


@pytest.mark.gpu
@pytest.mark.parametrize("dense_w", [False, True], ids=["sparse-W", "dense-W"])
def test_sparse_feedback_network_streams_packed_matrices(oracle_backend, hip_device, dense_w):
    """FeedbackNetwork whose coupling matrices -- inside and between the populations -- are fixed and mostly zeros: the
    populations' W are packed as for a single population, the edges between them too (`_packed_edges`), and `run` gives the
    bits of the dense run (here: of the oracle's dense arithmetic), recordings and states included."""
    r = np.random.default_rng(5)
    na, nb, T = (4096, 4100, 21) if dense_w else (1024, 2052, 37)

    def mat(rows, cols, gain, p=0.1):
        return (r.standard_normal((rows, cols)) * gain / np.sqrt(cols * p) * (r.uniform(size=(rows, cols)) < p)).astype(np.float32)

    # dense_w: dense recurrent matrices, sparse coupling BETWEEN the populations, everything by the defaults ("auto")
    Wa, Wb = (mat(na, na, 1.0, 1.0), mat(nb, nb, 1.0, 1.0)) if dense_w else (mat(na, na, 1.0), mat(nb, nb, 1.0))
    Wab, Wba = mat(nb, na, 0.6), mat(na, nb, 0.6)
    x = (r.standard_normal((T, na)) * 0.5).astype(np.float32)
    res = []
    for target in (dict(device="cpu", backend=oracle_backend), dict(device=hip_device)):
        net = FeedbackNetwork(1e-2, **target)
        for lab, W in (("a", Wa), ("b", Wb)):
            net.add_diffeq_node(lab, TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                                output_var="li_op/v", dtype=torch.float32, sparse_weights="auto" if dense_w else True)
        net.add_edge("a", "b", weights=Wab)
        net.add_edge("b", "a", weights=Wba, feedback=True)
        obs = net.run(x, sampling_steps=2, cutoff=3, verbose=False, enable_grad=False, record_vars=[("a", "li_op/v", False)])
        if target["device"] != "cpu":
            assert all((net.get_node(n)._pop.packed is None) == dense_w for n in "ab")
            assert all(getattr(net.get_edge(s_, t_), "_packed_cache", (None, None))[1] is not None for s_, t_ in (("a", "b"), ("b", "a")))
            assert all(float(net.get_node(n)._pop._edge_sum.abs().sum()) > 0 for n in "ab")
        res.append((obs.to_numpy("out"), obs.to_numpy(("a", "li_op/v")), {k_: v_.cpu().numpy() for k_, v_ in net.state.items()}))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    for k_ in res[0][2]:
        assert np.array_equal(res[0][2][k_], res[1][2][k_]), k_


@pytest.mark.gpu
def test_weights_edited_after_packing_are_repacked_when_the_matrices_move(hip_device, monkeypatch):
    """(advisor, r04) `node["weights"]` is live: an in-place edit AFTER the matrix was packed at construction must reach the
    coupled path -- also when the first run moves the matrices into one arena (`_consolidate`), which used to stamp the OLD
    packed form as current.  Packed run after the edit == dense run of the edited matrix, bit for bit."""
    monkeypatch.setattr(FeedbackNetwork, "ARENA_MIN_BYTES", 0)
    r = np.random.default_rng(11)
    n, T = 1024, 17
    mat = lambda rows, cols, p=0.1: (r.standard_normal((rows, cols)) / np.sqrt(cols * p) * (r.uniform(size=(rows, cols)) < p)).astype(np.float32)
    Wa, Wb, Wab, Wba = mat(n, n), mat(n, n), mat(n, n) * 0.6, mat(n, n) * 0.6
    x = (r.standard_normal((T, n)) * 0.5).astype(np.float32)
    outs = []
    for sparse in (True, False):
        net = FeedbackNetwork(1e-2, device=hip_device)
        for lab, W in (("a", Wa), ("b", Wb)):
            net.add_diffeq_node(lab, TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                                output_var="li_op/v", dtype=torch.float32, sparse_weights=sparse)
        net.add_edge("a", "b", weights=Wab)
        net.add_edge("b", "a", weights=Wba, feedback=True)
        pop = net.get_node("a")._pop
        assert (pop.packed is not None) == sparse
        with torch.no_grad():
            net.get_node("a")["weights"].mul_(1.7)        # after packing, before the first run
        outs.append(net.run(x, verbose=False, enable_grad=False).to_numpy("out"))
        assert net._arena is not None and pop.W.untyped_storage().data_ptr() == net._arena.untyped_storage().data_ptr()
    assert np.array_equal(outs[0], outs[1])
    assert np.abs(outs[0]).max() > 0


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_readout_only_fit_behind_coupled_populations_skips_the_adjoint(target, monkeypatch):
    """fit_bptt with only the read-out trained: no gradient has to pass into the populations, so they integrate without a
    graph (no rp_backward_coupled call, the windows of `run`); the read-out after three windows equals the one autograd over
    the per-step walk leaves."""
    bwd, fwd = [], []
    real_b, real_f = engine.backward_coupled, engine.forward_coupled
    monkeypatch.setattr(engine, "backward_coupled", lambda members: bwd.append(len(members)) or real_b(members))
    monkeypatch.setattr(engine, "forward_coupled", lambda members: fwd.append(len(members)) or real_f(members))
    r = np.random.default_rng(7)
    x, y = r.standard_normal((60, 3)).astype(np.float32), r.standard_normal((60, 2)).astype(np.float32) * 0.3
    res = []
    for stepwise in (True, False):
        bwd.clear(), fwd.clear()
        net = build_net(target, False, sizes=(16, 24), k=0)
        net.add_func_node("out", 2, "identity")
        edge = net.add_edge("b", "out", weights=r.standard_normal((2, 24)) * 0.3 if stepwise else res[0][2], train="gd")
        start = edge.weights.detach().cpu().numpy().copy()
        net._force_stepwise_adjoint = stepwise
        net._force_stepwise = stepwise
        obs = net.fit_bptt(x, y, optimizer="sgd", lr=0.05, update_steps=20, sampling_steps=1, verbose=False)
        assert bwd == [] and fwd == ([] if stepwise else [2] * 3)
        res.append((edge.weights.detach().cpu().numpy().copy(), obs.to_numpy("out"), start))
    assert np.abs(res[0][0] - res[1][0]).max() < 2e-5 and np.abs(res[0][1] - res[1][1]).max() < 2e-5
    assert np.abs(res[0][0] - res[0][2]).max() > 1e-3   # (the read-out did move)


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("pack_w", [False, True], ids=["edges", "edges-and-W"])
@pytest.mark.parametrize("sizes,out_vars,post", [((1024, 520), (0, 1), ()), ((260, 1028, 516), (1, 0, 1), ((0, 2),))])
def test_coupled_adjoint_with_packed_transposed_edges(target, oracle_backend, sizes, out_vars, post, pack_w):
    """Fixed sparse edges -- and, "edges-and-W", fixed sparse recurrent matrices -- in the adjoint: c = W_e^T gx (z = W^T u) by
    packed row dots of the transposed matrix (rp_bwd_member.edge_spt, rp_bwd.spt_*).
    Against float64 autograd over the walk (its own summation order, so a tolerance) -- and on the GPU additionally against
    the oracle's mirror of that order, bit for bit."""
    from rectipy_amd import sparse
    dev, T = target["device"], 17
    models = ("li_tanh",) * len(sizes)
    gen = torch.Generator().manual_seed(9)
    x = (torch.randn(T, sizes[0], generator=gen) * 0.5)
    g_out = [torch.randn(T, n, generator=gen) for n in sizes]
    g_vend = [torch.randn(n, generator=gen) for n in sizes]

    def run(device, backend):
        pops, edges = sparse_circuit(sizes, models, 5, device, backend, density=0.15)
        if pack_w:   # the members' own fixed sparse W: z = W^T u by the packed rows of W^T (rp_bwd.spt_*)
            for p in pops:
                p.pack_weights()
        packed_t = {key: sparse.pack_rows(We.T.contiguous()) for key, We in edges.items()}
        res, _ = coupled_adjoint(pops, edges, out_vars, x.to(device), T, g_out, g_vend, device, post=post, packed_t=packed_t)
        return pops, edges, res

    pops, edges, res = run(dev, target["backend"])
    d = lambda t: t.detach().cpu().double().clone().requires_grad_(True)
    leaves = dict(v0=[d(p.v) for p in pops], W=[d(p.W) for p in pops], tau=[d(p.params["tau"]) for p in pops],
                  k=[d(p.params["k"]) for p in pops], eta=[d(p.params["eta"]) for p in pops],
                  edges={key: d(We) for key, We in edges.items()}, x=d(x))
    # (the circuit's states moved during the forward: rebuild it for the float64 walk)
    pops0, _ = sparse_circuit(sizes, models, 5, "cpu", oracle_backend, density=0.15)
    leaves["v0"] = [d(p.v) for p in pops0]
    outs, v_end = torch_walk_rate(pops0, edges, out_vars, x, T, leaves, post)
    loss = sum((o * g.double()).sum() for o, g in zip(outs, g_out)) + sum((v * g.double()).sum() for v, g in zip(v_end, g_vend))
    loss.backward()
    for m in range(len(sizes)):
        for got, want, what in ((res[m]["lam"], leaves["v0"][m].grad, "d/dv0"), (res[m]["W"], leaves["W"][m].grad, "dW"),
                                (res[m]["tau"], leaves["tau"][m].grad, "dtau")):
            scale = float(want.abs().max()) + 1e-12
            assert float((got.detach().cpu().double() - want).abs().max()) <= 3e-4 * scale, f"member {m} {what}"
    if dev != "cpu":
        _, _, ref = run("cpu", oracle_backend)
        for m in range(len(sizes)):
            for key in ("lam", "u", "gx", "g_out", "tau", "k", "eta"):
                assert_same(ref[m][key], res[m][key], f"member {m} {key}")


@pytest.mark.gpu
def test_fit_with_fixed_sparse_edges_packs_them_in_both_directions(hip_device, monkeypatch):
    """Trained dense recurrent matrices, FIXED sparse edges between the populations: the forward streams the packed edges
    (rp_run.edge_sp), the adjoint the packed transposed ones (rp_bwd_member.edge_spt); the parameters after three SGD windows
    equal those with the dense edges to float32 rounding (different summation order of the edge products)."""
    r = np.random.default_rng(3)
    na, nb, T = 2048, 1028, 60
    mat = lambda rows, cols, g, p: (r.standard_normal((rows, cols)) * g / np.sqrt(cols * p) * (r.uniform(size=(rows, cols)) < p)).astype(np.float32)
    Wa, Wb, Eab, Eba = mat(na, na, 0.9, 1.0), mat(nb, nb, 0.9, 1.0), mat(nb, na, 0.5, 0.05), mat(na, nb, 0.5, 0.05)
    Wout = (r.standard_normal((2, nb)) / np.sqrt(nb)).astype(np.float32)
    x, y = (r.standard_normal((T, na)) * 0.3).astype(np.float32), (r.standard_normal((T, 2)) * 0.2).astype(np.float32)
    res = []
    for packed in (False, True):
        net = FeedbackNetwork(1e-2, device=hip_device)
        net.PACKED_EDGE_MIN_ROWS = 0
        if not packed:
            net.PACKED_EDGE_MAX_DENSITY = 0.0
        for lab, W in (("a", Wa), ("b", Wb)):
            net.add_diffeq_node(lab, TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                                output_var="li_op/v", dtype=torch.float32, train_params=["weights", "li_op/tau"])
        net.add_func_node("out", 2, "identity")
        net.add_edge("a", "b", weights=Eab)
        net.add_edge("b", "out", weights=Wout, train="gd")
        net.add_edge("b", "a", weights=Eba, feedback=True)
        obs = net.fit_bptt(x, y, optimizer="sgd", lr=0.02, update_steps=20, sampling_steps=1, verbose=False)
        for s_, t_ in (("a", "b"), ("b", "a")):
            e_ = net.get_edge(s_, t_)
            assert (getattr(e_, "_packed_cache", (None, None))[1] is not None) == packed
            assert (getattr(e_, "_packed_cache_t", (None, None))[1] is not None) == packed
        res.append(([p_.detach().cpu().numpy().copy() for p_ in net.parameters()], obs.to_numpy("out")))
    for a_, b_ in zip(res[0][0], res[1][0]):
        assert np.abs(a_ - b_).max() <= 2e-5 * max(1.0, np.abs(a_).max())
    assert np.abs(res[0][1] - res[1][1]).max() < 2e-5 * max(1.0, np.abs(res[0][1]).max())


@pytest.mark.gpu
def test_packed_edges_with_a_source_wider_than_the_lds_window(hip_device):
    """An edge whose SOURCE holds more than 32768 neurons is packed in the windowed form (the source vector staged window by
    window, sparse.pack_rows_windowed), as a population's own matrix of that width: a pair of 4096 and 33024 neurons with every
    matrix at p = 0.05, packed run against the same network streamed densely on the same device, bit for bit."""
    from rectipy_amd import sparse
    na, nb, T, p = 4096, 33024, 6, 0.05
    gen = torch.Generator(device=hip_device).manual_seed(2)

    def mat(rows, cols, g):
        W = torch.randn(rows, cols, device=hip_device, generator=gen) * (g / (cols * p) ** 0.5)
        return W * (torch.rand(rows, cols, device=hip_device, generator=gen) < p)

    Wa, Wb, Eab, Eba = mat(na, na, 1.0), mat(nb, nb, 1.0), mat(nb, na, 0.5), mat(na, nb, 0.5)
    x = torch.randn(T, na, device=hip_device, generator=gen) * 0.3
    res = []
    for packed in (True, False):
        pops = [Population("li_tanh", W.clone(), dict(tau=2.0, k=1.0, eta=0.0), 1e-2, hip_device) for W in (Wa, Wb)]
        pk = {}
        if packed:
            for pop in pops:
                pop.pack_weights()
            pk = {(0, 1): sparse.pack_rows_blocked(Eab, block_rows=pops[1]._block_rows()) if pops[1]._block_rows() else sparse.pack_rows(Eab),
                  (1, 0): sparse.pack_rows_windowed(Eba)}
            assert pk[(1, 0)].row_ids is not None and pops[1].packed.row_ids is not None
        trajs = [torch.zeros(T, pop.rows, device=hip_device) for pop in pops]
        hist = [[torch.zeros(pop.rows, device=hip_device) for _ in range(3)] for pop in pops]
        forward_coupled([
            dict(pop=pops[0], run=dict(steps=T, x=x, traj=trajs[0], hist=hist[0]), edges=[(Eba, 1, _abi.RP_EDGE_LIVE, pk.get((1, 0)))]),
            dict(pop=pops[1], run=dict(steps=T, traj=trajs[1], hist=hist[1]), edges=[(Eab, 0, _abi.RP_EDGE_OUT, pk.get((0, 1)))])])
        res.append([t.clone() for t in trajs] + [pop.v.clone() for pop in pops])
        del pops, pk
        torch.cuda.empty_cache()
    for a, b in zip(*res):
        assert_same(a, b, "packed vs dense")


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("kind", ["plain", "momentum"])
def test_plain_sgd_update_inside_the_gradient_gemm(target, kind, monkeypatch):
    """fit_bptt under plain SGD applies the update of the trained matrices in the gradient GEMM (W += -lr dW, Network._fused_sgd_plan)
    in every window but the call's last complete one; with RECTIPY_AMD_FUSED_SGD=0 the optimiser makes its passes.  Same parameters
    (the rounding of lr * dW differs in the last bit), the same predictions, and the last window's gradient on every parameter
    either way.  A masked edge keeps the optimiser's route, and so does everything under an optimiser with a state (momentum)."""
    from rectipy_amd.engine import Population
    accum = []
    real = Population.outer_accum
    monkeypatch.setattr(Population, "outer_accum", lambda self, *a: accum.append("W") or real(self, *a))
    r = np.random.default_rng(3)
    sizes, T, win = dict(a=24, b=40), 67, 12      # five complete windows and a ragged tail
    x, y = r.standard_normal((T, 24)).astype(np.float32) * 0.5, r.standard_normal((T, 3)).astype(np.float32) * 0.3
    init = {k_: r.standard_normal(s_) * 0.3 for k_, s_ in dict(a=(24, 24), b=(40, 40), ab=(40, 24), ba=(24, 40), bb=(24, 40), out=(3, 40)).items()}
    mask = (r.uniform(size=(24, 40)) < 0.5).astype(np.float32)
    res = []
    for env in ("1", "0"):
        monkeypatch.setenv("RECTIPY_AMD_FUSED_SGD", env)
        accum.clear()
        net = FeedbackNetwork(1e-2, **target)
        for lab in "ab":
            net.add_diffeq_node(lab, TANH, weights=init[lab], source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                                output_var="li_op/v", dtype=torch.float32, train_params=["weights", "li_op/tau"],
                                node_vars={"all/li_op/tau": r.uniform(0.5, 2, sizes[lab]) * 0 + 1.0})
        net.add_func_node("out", 3, "identity")
        edges = [net.add_edge("a", "b", weights=init["ab"], train="gd"), net.add_edge("b", "a", weights=init["ba"], feedback=True, train="gd"),
                 net.add_edge("b", "out", weights=init["out"], train="gd")]
        backend = net.get_node("a")._pop.backend
        real_rect = type(backend).outer_accum_rect
        monkeypatch.setattr(type(backend), "outer_accum_rect", lambda self, *a: accum.append("E") or real_rect(self, *a))
        obs = net.fit_bptt(x, y, optimizer="sgd", lr=5.0, update_steps=win, sampling_steps=1, verbose=False,
                           optimizer_kwargs=dict(momentum=0.5) if kind == "momentum" else None)
        monkeypatch.setattr(type(backend), "outer_accum_rect", real_rect)
        fused_windows = 4 if env == "1" and kind == "plain" else 0     # (five complete windows: all but the last)
        assert sorted(accum) == ["E"] * (2 * fused_windows) + ["W"] * (2 * fused_windows)
        params = [net.get_node(lab)[k_] for lab in "ab" for k_ in ("weights", "li_op/tau")] + [e.weights for e in edges]
        assert all(p.grad is not None for p in params)
        res.append(([p.detach().cpu().numpy().copy() for p in params], [p.grad.detach().cpu().numpy().copy() for p in params],
                    obs.to_numpy("out"), obs.to_numpy("loss")))
    for (p1, p0, g1, g0) in zip(res[0][0], res[1][0], res[0][1], res[1][1]):
        assert np.abs(p1 - p0).max() <= 1e-6 and np.abs(g1 - g0).max() <= 1e-4 * np.abs(g0).max()
    assert np.abs(res[0][2] - res[1][2]).max() < 1e-5 and np.abs(res[0][3] - res[1][3]).max() < 1e-5
    for i, key in ((0, "a"), (2, "b"), (4, "ab"), (5, "ba")):
        assert np.abs(res[0][0][i] - init[key]).max() > 5e-5, key     # (the matrices did move: fifty times the tolerance and more)


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("shape", ["one-population", "two-in-a-row"])
def test_plain_sgd_update_inside_the_gradient_gemm_single_population_windows(target, shape, monkeypatch):
    """The same for Network.fit_bptt over a population's own autograd window (_RateWindow; one population behind a read-out, and two
    populations in a row -- no cycle): every window but the last complete one updates the trained W in the gradient GEMM."""
    from rectipy_amd import Network
    from rectipy_amd.engine import Population
    accum = []
    real = Population.outer_accum
    monkeypatch.setattr(Population, "outer_accum", lambda self, *a: accum.append("W") or real(self, *a))
    r = np.random.default_rng(5)
    n, T, win = 48, 45, 10      # four complete windows and a ragged tail
    labs = "a" if shape == "one-population" else "ab"
    x, y = r.standard_normal((T, n)).astype(np.float32) * 0.5, r.standard_normal((T, 3)).astype(np.float32) * 0.3
    init = {lab: r.standard_normal((n, n)) * 0.3 for lab in labs}
    W_out, W_ab = r.standard_normal((3, n)) * 0.3, r.standard_normal((n, n)) * 0.2
    res = []
    for env in ("1", "0"):
        monkeypatch.setenv("RECTIPY_AMD_FUSED_SGD", env)
        accum.clear()
        net = Network(1e-2, **target)
        for lab in labs:
            net.add_diffeq_node(lab, TANH, weights=init[lab], source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                                output_var="li_op/v", dtype=torch.float32, train_params=["weights", "li_op/k"])
        net.add_func_node("out", 3, "identity")
        edges = [net.add_edge(labs[-1], "out", weights=W_out, train="gd")]
        if len(labs) == 2:
            edges.append(net.add_edge("a", "b", weights=W_ab, train="gd"))
        obs = net.fit_bptt(x, y, optimizer="sgd", lr=5.0, update_steps=win, sampling_steps=1, verbose=False)
        assert accum == ["W"] * (len(labs) * (3 if env == "1" else 0))
        params = [net.get_node(lab)[k_] for lab in labs for k_ in ("weights", "li_op/k")] + [e.weights for e in edges]
        assert all(p.grad is not None for p in params)
        res.append(([p.detach().cpu().numpy().copy() for p in params], [p.grad.detach().cpu().numpy().copy() for p in params],
                    obs.to_numpy("out"), obs.to_numpy("loss")))
    for (p1, p0, g1, g0) in zip(res[0][0], res[1][0], res[0][1], res[1][1]):
        assert np.abs(p1 - p0).max() <= 1e-6 and np.abs(g1 - g0).max() <= 1e-4 * np.abs(g0).max()
    assert np.abs(res[0][2] - res[1][2]).max() < 1e-5 and np.abs(res[0][3] - res[1][3]).max() < 1e-5
    assert np.abs(res[0][0][0] - init["a"]).max() > 5e-5
