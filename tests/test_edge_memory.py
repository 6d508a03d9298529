"""Edges with memory and activation nodes INSIDE the library calls (ABI v17): rp_edge_mem / rp_edge_window -- the delay buffer
of LinearMemory, the recursive filter of LinearFilter, both for LinearMemoryFilter (reference rectipy/edges.py:68-147) -- and
RP_MODEL_INSTANT members / identity edges of rp_forward_coupled (reference InstantNode, rectipy/nodes.py:14-51, evaluated per
step by rectipy/network.py:962-977, 1322-1357).

CPU: the oracle's stages against the package's torch restatement of the reference's edge classes (which
tests/test_reference_fuzz.py compares with the reference's own classes step by step), and a coupled circuit with an
activation node, identity edges and edges with memory against an independent float64 walk.  GPU: the HIP library against the
oracle, bit for bit."""
import numpy as np
import pytest
import torch

from rectipy_amd import _abi
from rectipy_amd.edge_variants import LinearFilter, LinearMemory, LinearMemoryFilter
from rectipy_amd.engine import EdgeMemory, InstantPopulation, Population, forward_coupled

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


def col_chan_of(delays: np.ndarray) -> np.ndarray:
    """buffer[:, delays] = x: column c receives x[j] for the LAST j with delays[j] == c (the CPU's indexed assignment)."""
    cc = -np.ones(int(delays.max()) + 1, dtype=np.int32)
    for j, d in enumerate(delays):
        cc[d] = j
    return cc


def make_stage(kind, n, delays, F, dev, backend):
    if kind == "filter":
        return EdgeMemory("filter", n, dev, backend=backend, F=F)
    cc = col_chan_of(delays)
    return EdgeMemory(kind, n, dev, backend=backend, F=F if kind == "delay_filter" else None, depth=len(cc), col_chan=cc)


STAGES = [("filter", 5, None), ("filter", 8, None), ("filter", 64, None), ("filter", 70, None), ("filter", 260, None),
          ("delay", 6, (1, 4)), ("delay", 300, (0, 2, 9)), ("delay", 64, (3,)),
          ("delay_filter", 6, (1, 4)), ("delay_filter", 8, (0, 5)), ("delay_filter", 64, (2, 3, 7)),
          ("delay_filter", 70, (2, 3, 7)), ("delay_filter", 200, (1, 6)), ("delay_filter", 48, tuple(range(0, 180, 7)))]


def stage_case(kind, n, choices, seed=0):
    gen = torch.Generator().manual_seed(seed + n)
    rng = np.random.default_rng(seed + n)
    F = torch.randn(n, n, generator=gen) * 0.8 / n ** 0.5
    delays = rng.choice(choices, size=n) if choices else None
    if delays is not None:
        delays[rng.integers(n)] = max(choices)
    return F, delays


@pytest.mark.parametrize("kind,n,choices", STAGES)
def test_oracle_stage_is_the_edge_classes_memory(kind, n, choices, oracle_backend):
    """T steps of the oracle's stage == T calls of the torch edge class's `_advance` (the reference's forward without the
    projection): pure selection for the delay buffer (exact), the filters up to the rounding order of F v."""
    T = 17
    F, delays = stage_case(kind, n, choices)
    X = torch.randn(T, n, generator=torch.Generator().manual_seed(3))
    W = np.zeros((2, n), dtype=np.float32)
    if kind == "filter":
        edge = LinearFilter(n, 2, F.numpy(), weights=W, dtype=torch.float32)
        edge.y = torch.randn(n, generator=torch.Generator().manual_seed(4))
        start = edge.y.clone()
    elif kind == "delay":
        edge = LinearMemory(n, 2, delays, weights=W, dtype=torch.float32)
        edge.buffer = torch.randn(1, edge.buffer.shape[1], generator=torch.Generator().manual_seed(4)).expand(n, -1).contiguous()
        start = edge.buffer[0].clone()
    else:
        edge = LinearMemoryFilter(n, 2, delays, F.numpy(), weights=W, dtype=torch.float32)
        edge.buffer = torch.randn(edge.buffer.shape, generator=torch.Generator().manual_seed(4))
        start = edge.buffer.T.clone()
    ref = torch.stack([edge._advance(X[t]).clone() for t in range(T)])
    mem = make_stage(kind, n, delays, F, "cpu", oracle_backend)
    mem.load(start)
    U = mem.window(X[:9])
    U = torch.cat([U, mem.window(X[9:])])          # two calls: the state carries over
    end = {"filter": lambda: edge.y, "delay": lambda: edge.buffer[0], "delay_filter": lambda: edge.buffer.T}[kind]()
    tol = 0.0 if kind == "delay" else 2e-6 * float(ref.abs().max())
    assert float((U - ref).abs().max()) <= tol
    assert float((mem.current - end).abs().max()) <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,choices", STAGES + [("filter", 1028, None), ("filter", 4100, None), ("delay_filter", 516, (0, 3))])
def test_hip_edge_window_bit_exact(kind, n, choices, oracle_backend, hip_device):
    """rp_edge_window on the GPU == the oracle, bit for bit: window call, then single steps (the per-step form inside the
    coupled loop shares the kernels), narrow (one workgroup, chain order) and wide (row-dot order, 16-byte and 4-byte paths)."""
    T = 11
    F, delays = stage_case(kind, n, choices)
    X = torch.randn(T, n, generator=torch.Generator().manual_seed(5))
    a = make_stage(kind, n, delays, F, "cpu", oracle_backend)
    b = make_stage(kind, n, delays, F, hip_device, None)
    start = torch.randn(a.state[0].shape, generator=torch.Generator().manual_seed(6))
    if kind == "delay":
        pass
    a.load(start)
    b.load(start)
    Ua = torch.cat([a.window(X[:7])] + [a.window(X[t:t + 1]) for t in range(7, T)])
    Xd = X.to(hip_device)
    Ub = torch.cat([b.window(Xd[:7])] + [b.window(Xd[t:t + 1]) for t in range(7, T)])
    torch.cuda.synchronize()
    assert_same(Ua, Ub, f"{kind} n={n}: outputs")
    assert_same(a.current, b.current, f"{kind} n={n}: state")


# ---- a coupled circuit with an activation node, identity edges and edges with memory ---------------------------------------
def build_circuit(dev, backend, sizes=(24, 16, 20), seed=0, dt=1e-2):
    """a (tanh, n_a) -> f (activation node `tanh`, n_f) -> b (tanh, n_b); a -> b directly through a DELAY edge; feedback b -> a
    through a FILTER edge; and b's third input through a partial-sum member g (identity activation, width n_b) that adds two
    plain edges from a and f and hands the sum on through an identity edge."""
    na, nf, nb = sizes
    gen = torch.Generator().manual_seed(seed)
    rnd = lambda *shape, scale=1.0: (torch.randn(*shape, generator=gen) * scale)

    def pop(n):
        return Population("li_tanh", rnd(n, n, scale=1.2 / n ** 0.5), dict(tau=torch.rand(n, generator=gen) * 2 + 0.5,
                          k=torch.rand(n, generator=gen) + 0.5, eta=rnd(n, scale=0.1)), dt, dev, v0=rnd(n, scale=0.5), backend=backend)

    a, b = pop(na), pop(nb)
    f = InstantPopulation(nf, "tanh", dev, backend=backend)
    g = InstantPopulation(nb, "identity", dev, backend=backend)
    W = dict(af=rnd(nf, na, scale=0.8 / na ** 0.5), fb=rnd(nb, nf, scale=0.8 / nf ** 0.5), ab=rnd(nb, na, scale=0.5 / na ** 0.5),
             ba=rnd(na, nb, scale=0.5 / nb ** 0.5), ag=rnd(nb, na, scale=0.4 / na ** 0.5), fg=rnd(nb, nf, scale=0.4 / nf ** 0.5))
    W = {k: v.to(dev) for k, v in W.items()}
    delays = np.asarray([(3 * j) % 4 + 1 for j in range(na)])
    mem_ab = make_stage("delay", na, delays, None, dev, backend)
    mem_ba = make_stage("filter", nb, None, rnd(nb, nb, scale=0.6 / nb ** 0.5), dev, backend)
    return dict(a=a, f=f, g=g, b=b, W=W, delays=delays, mem_ab=mem_ab, mem_ba=mem_ba)


def run_built(C, x, T, split, dev):
    a, f, g, b, W = C["a"], C["f"], C["g"], C["b"], C["W"]
    order = [a, f, g, b]
    trajs = [torch.zeros(T, p.rows, device=dev) for p in order]
    bounds = np.linspace(0, T, split + 1).astype(int)
    OUT, LIVE = _abi.RP_EDGE_OUT, _abi.RP_EDGE_LIVE
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        run = lambda m: dict(steps=int(hi - lo), step0=int(lo), out_var=_abi.RP_OUT_SOURCE, traj=trajs[m][lo:hi])
        members = [
            dict(pop=a, run=dict(run(0), x=x[lo:hi]), edges=[(W["ba"], 3, LIVE, None, C["mem_ba"])]),
            dict(pop=f, run=run(1), edges=[(W["af"], 0, OUT)]),
            dict(pop=g, run=run(2), edges=[(W["ag"], 0, OUT), (W["fg"], 1, OUT)]),
            dict(pop=b, run=run(3), edges=[(W["fb"], 1, OUT), (W["ab"], 0, OUT, None, C["mem_ab"]), (None, 2, OUT)]),
        ]
        forward_coupled(members)
    return trajs + [a.v, b.v, C["mem_ab"].current, C["mem_ba"].current]


def float64_walk(C, x, T):
    f64 = lambda t: t.detach().cpu().numpy().astype(np.float64)
    a, b, W = C["a"], C["b"], {k: f64(v) for k, v in C["W"].items()}
    S = {k: dict(v=f64(p.v), W=f64(p.W), par={q: f64(t) for q, t in p.params.items()}) for k, p in (("a", a), ("b", b))}
    cc = col_chan_of(C["delays"])
    ring = f64(C["mem_ab"].current)
    y = f64(C["mem_ba"].current)
    F = f64(C["mem_ba"].F)
    outs = dict(a=[], f=[], g=[], b=[])
    dt = a.dt
    for t in range(T):
        ra, rb = np.tanh(S["a"]["v"]), np.tanh(S["b"]["v"])
        # a: input x + feedback from b (its output as it stands: b has not been stepped) through the filter edge
        y = F @ y + rb
        Ia = f64(x[t]) + W["ba"] @ y
        fo = np.tanh(W["af"] @ ra)
        go = W["ag"] @ ra + W["fg"] @ fo
        ring = np.asarray([ra[cc[c]] if cc[c] >= 0 else ring[(c + 1) % len(ring)] for c in range(len(ring))])
        Ib = (W["fb"] @ fo + W["ab"] @ (ring[0] * np.ones_like(ra))) + go
        for k, r, I in (("a", ra, Ia), ("b", rb, Ib)):
            P = S[k]
            P["v"] = P["v"] + dt * (-P["v"] / P["par"]["tau"] + P["par"]["k"] * (P["W"] @ r) + I + P["par"]["eta"])
        for k, o in (("a", ra), ("f", fo), ("g", go), ("b", rb)):
            outs[k].append(o)
    return [np.asarray(outs[k]) for k in "afgb"]


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("sizes", [(24, 16, 20), (260, 72, 132), (17, 9, 30)])
def test_coupled_circuit_with_activation_node_identity_and_memory_edges(target, sizes, oracle_backend):
    dev = target["device"]
    T = 40
    x = torch.randn(T, sizes[0], generator=torch.Generator().manual_seed(9)) * 0.3
    C = build_circuit(dev, target["backend"], sizes)
    ref = float64_walk(C, x, T)
    got = run_built(C, x.to(dev), T, 3, dev)
    for r, t_, name in zip(ref, got, "afgb"):
        assert float(np.abs(t_.cpu().numpy() - r).max()) < 2e-5, name
    if dev != "cpu":  # the same circuit on the oracle: bit for bit
        Co = build_circuit("cpu", oracle_backend, sizes)
        want = run_built(Co, x, T, 3, "cpu")
        for w, g_, name in zip(want, got, ["a", "f", "g", "b", "v_a", "v_b", "delay row", "filter state"]):
            assert_same(w, g_, name)
    # one call for the whole window == three calls
    C1 = build_circuit(dev, target["backend"], sizes)
    once = run_built(C1, x.to(dev), T, 1, dev)
    for w, g_ in zip(once, got):
        assert_same(w, g_, "split windows")


# ---- the adjoint of such a circuit (rp_backward_coupled, ABI v18) -----------------------------------------------------------
def adjoint_built(C, x, T, g_out, g_end, dev):
    """a -> f (tanh node) -> b, a -> b through the DELAY edge, feedback b -> a through the FILTER edge: the window stored, then ONE
    rp_backward_coupled call.  Returns per member its gradient tensors and, per edge, the weight gradient gx_target^T (what the
    edge's projection read)."""
    import types
    from rectipy_amd.engine import backward_coupled
    a, f, b, W = C["a"], C["f"], C["b"], C["W"]
    f32 = dict(dtype=torch.float32, device=dev)
    order = [a, f, b]
    OUT, LIVE = _abi.RP_EDGE_OUT, _abi.RP_EDGE_LIVE
    saved = [dict(v=torch.zeros(T, p.rows, **f32), r=torch.zeros(T, p.rows, **f32), acc=torch.zeros(T, p.rows, **f32)) for p in order]
    for mem, n in ((C["mem_ab"], a.rows), (C["mem_ba"], b.rows)):
        mem.save, mem.save_row0 = torch.zeros(T, (n + 3) // 4 * 4, **f32), 0
    run = lambda m: dict(steps=T, step0=0, out_var=_abi.RP_OUT_SOURCE, save_v=saved[m]["v"], save_src=saved[m]["r"], save_acc=saved[m]["acc"])
    fwd = [dict(pop=a, run=dict(run(0), x=x), edges=[(W["ba"], 2, LIVE, None, C["mem_ba"])]),
           dict(pop=f, run=dict(steps=T, step0=0, out_var=_abi.RP_OUT_SOURCE, traj=saved[1]["r"]), edges=[(W["af"], 0, OUT)]),
           dict(pop=b, run=run(2), edges=[(W["fb"], 1, OUT), (W["ab"], 0, OUT, None, C["mem_ab"])])]
    forward_coupled(fwd)
    res, specs = [], []
    for m, p in enumerate(order):
        R = dict(g_out=g_out[m].to(**f32).clone(), gx=torch.zeros(T, p.rows, **f32))
        if m == 1:
            kw = dict(instant=True, traj_r=saved[1]["r"], grad_out=R["g_out"], grad_x=R["gx"])
        else:
            R.update(lam=g_end[m].to(**f32).clone(), u=torch.zeros(T, p.rows, **f32), tau=torch.zeros(p.rows, **f32),
                     k=torch.zeros(p.rows, **f32), eta=torch.zeros(p.rows, **f32))
            kw = dict(traj_v=saved[m]["v"], traj_r=saved[m]["r"], grad_out=R["g_out"], grad_var=_abi.RP_OUT_SOURCE, lam=R["lam"],
                      u_traj=R["u"], traj_acc=saved[m]["acc"], grad_x=R["gx"], grad_tau=R["tau"], grad_k=R["k"], grad_eta=R["eta"])
        res.append(R)
        specs.append(dict(pop=p, bwd=kw))
    d = int(C["delays"].min())
    delay_obj = types.SimpleNamespace(_cols=[d], _src=[int(np.nonzero(C["delays"] == d)[0].max())])
    filter_obj = types.SimpleNamespace(filter=C["mem_ba"].F)
    specs[0]["edges"] = [dict(W=W["ba"], source=2, mode=LIVE, memory="filter", edge=filter_obj)]
    specs[1]["edges"] = [dict(W=W["af"], source=0, mode=OUT)]
    specs[2]["edges"] = [dict(W=W["fb"], source=1, mode=OUT), dict(W=W["ab"], source=0, mode=OUT, memory="delay", edge=delay_obj)]
    keep = backward_coupled(specs)
    del keep
    gW = dict(ba=res[0]["gx"].T @ C["mem_ba"].save[:, :b.rows], af=res[1]["gx"].T @ saved[0]["r"], fb=res[2]["gx"].T @ saved[1]["r"],
              ab=res[2]["gx"].T @ C["mem_ab"].save[:, :a.rows])
    return res, gW


def float64_adjoint(C, x, T, g_out, g_end):
    """torch float64 restatement of the same window, differentiated by autograd: (gradient of x, of the start states, of the edge
    matrices) for the loss sum_m <g_out_m, out_m> + sum_pops <g_end, v_end>."""
    f64 = lambda t: t.detach().cpu().to(torch.float64)
    a, b = C["a"], C["b"]
    W = {k: f64(v).requires_grad_(True) for k, v in C["W"].items() if k in ("ba", "af", "fb", "ab")}
    xs = f64(x).requires_grad_(True)
    v0 = dict(a=f64(a.v).requires_grad_(True), b=f64(b.v).requires_grad_(True))
    par = {k: dict(W=f64(p.W), **{q: f64(t) for q, t in p.params.items()}) for k, p in (("a", a), ("b", b))}
    cc = col_chan_of(C["delays"])
    ring = [f64(C["mem_ab"].current)[c] for c in range(len(cc))]
    y, F = f64(C["mem_ba"].current), f64(C["mem_ba"].F)
    v, dt = dict(v0), a.dt
    loss = 0.0
    for t in range(T):
        ra, rb = torch.tanh(v["a"]), torch.tanh(v["b"])
        y = F @ y + rb
        Ia = xs[t] + W["ba"] @ y
        fo = torch.tanh(W["af"] @ ra)
        ring = [ra[cc[c]] if cc[c] >= 0 else ring[(c + 1) % len(ring)] for c in range(len(ring))]
        Ib = W["fb"] @ fo + W["ab"] @ (ring[0] * torch.ones_like(ra))
        for k, r, I in (("a", ra, Ia), ("b", rb, Ib)):
            P = par[k]
            v[k] = v[k] + dt * (-v[k] / P["tau"] + P["k"] * (P["W"] @ r) + I + P["eta"])
        loss = loss + (f64(g_out[0][t]) * ra).sum() + (f64(g_out[1][t]) * fo).sum() + (f64(g_out[2][t]) * rb).sum()
    loss = loss + (f64(g_end[0]) * v["a"]).sum() + (f64(g_end[2]) * v["b"]).sum()
    loss.backward()
    return dict(x=xs.grad, va=v0["a"].grad, vb=v0["b"].grad, **{k: w.grad for k, w in W.items()})


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("sizes", [(24, 16, 20), (260, 72, 132), (17, 9, 30), (1028, 520, 2052), (4100, 64, 1024)])
def test_coupled_adjoint_through_activation_node_and_memory_edges(target, sizes, oracle_backend):
    """(r05, ABI v18) rp_backward_coupled through an activation-node member (gx = g_out * act'(out)), a FILTER edge on the feedback
    edge (ly(t) = W_e^T gx(t) + F^T ly(t + 1)) and a DELAY edge (the total of W_e^T gx(t) into one element, d steps earlier):
    against float64 autograd of the same window; on the GPU also bit for bit against the oracle."""
    dev = target["device"]
    T = 23
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(T, sizes[0], generator=gen) * 0.3
    order = (sizes[0], sizes[1], sizes[2])
    g_out = [torch.randn(T, n, generator=gen) for n in order]
    g_end = [torch.randn(n, generator=gen) for n in order]
    C = build_circuit(dev, target["backend"], sizes)
    starts = [torch.randn(mem.current.shape, generator=gen) * 0.3 for mem in (C["mem_ab"], C["mem_ba"])]
    for mem, src in zip((C["mem_ab"], C["mem_ba"]), starts):  # (a start state that is not zero)
        mem.load(src)
    want = float64_adjoint(C, x, T, g_out, g_end)
    res, gW = adjoint_built(C, x.to(dev), T, g_out, g_end, dev)
    close = lambda got, ref, what: np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5 * max(1.0, float(ref.abs().max())), err_msg=what)
    close(res[0]["gx"], want["x"], "grad x")
    close(res[0]["lam"], want["va"], "grad v_a(0)")
    close(res[2]["lam"], want["vb"], "grad v_b(0)")
    for k in ("ba", "af", "fb", "ab"):
        close(gW[k], want[k], f"grad W[{k}]")
    if dev != "cpu":
        Co = build_circuit("cpu", oracle_backend, sizes)
        for mem, src in ((Co["mem_ab"], starts[0]), (Co["mem_ba"], starts[1])):
            mem.load(src)
        ref, _ = adjoint_built(Co, x, T, g_out, g_end, "cpu")
        for m in range(3):
            for key in res[m]:
                assert_same(ref[m][key], res[m][key], f"member {m} {key}")
