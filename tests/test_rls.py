"""Recursive-least-squares readout training: front end + kernels against the reference's fit_rls (golden G14) and
HIP vs oracle bit for bit."""
import os

import numpy as np
import pytest
import torch

from rectipy_amd import Network
from rectipy_amd.rls import RLS

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
TARGETS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture
def target(request, oracle_backend):
    if request.param == "oracle":
        return dict(device="cpu", backend=oracle_backend)
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return dict(device="cuda:0", backend=None)


def build(g, target):
    net = Network(float(g["dt"]), **target)
    net.add_diffeq_node("rnn", TANH, weights=g["W"], source_var="tanh_op/r", target_var="li_op/r_in",
                        input_var="li_op/I_ext", output_var="li_op/v", dtype=torch.float32,
                        node_vars={"all/li_op/tau": g["tau"], "all/li_op/k": g["k"], "all/li_op/eta": g["eta"],
                                   "all/li_op/v": g["y0"]})
    net.add_func_node("out", g["W0"].shape[0], "identity")
    edge = net.add_edge("rnn", "out", weights=g["W0"].copy(), train="rls", beta=float(g["beta"]), alpha=float(g["alpha"]))
    return net, edge


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("stride", [1, 4])
@pytest.mark.parametrize("tiled", [False, True, "symmetric"], ids=["row-per-wave", "tiled", "symmetric"])
def test_g14_fit_rls_matches_reference(target, stride, tiled):
    g = dict(np.load(os.path.join(GOLD, "g14_rls.npz")))
    net, edge = build(g, target)
    edge.symmetric = tiled == "symmetric" and g["W"].shape[0] % 4 == 0
    edge.tiled = bool(tiled) and g["W"].shape[0] % 4 == 0
    assert isinstance(edge, RLS) and abs(edge.beta - 1 / 0.95) < 1e-12
    obs = net.fit_rls(g["x"], g["target"], update_steps=stride, sampling_steps=1, verbose=False)
    key = f"s{stride}_f64"
    scale = np.abs(g["W_after_" + key]).max()
    assert np.abs(edge.weights.cpu().numpy() - g["W_after_" + key]).max() < 2e-4 * max(1.0, scale)
    assert np.abs(edge.P.cpu().numpy() - g["P_after_" + key]).max() < 2e-4 * np.abs(g["P_after_" + key]).max()
    assert np.abs(obs.to_numpy("out") - g["pred_" + key]).max() < 2e-4
    ref_loss = g["loss_" + key]
    assert np.abs(obs.to_numpy("loss") - ref_loss).max() < 2e-4 * max(1.0, ref_loss.max())
    assert np.abs(net.state["rnn"].cpu().numpy() - g["state_" + key]).max() < 2e-5
    # per-step API (drop-in loops calling edge.update) agrees with the fused sweep
    net2, edge2 = build(g, target)
    net2.compile()
    x = torch.tensor(g["x"], dtype=torch.float32, device=target["device"])
    y = torch.tensor(g["target"], dtype=torch.float32, device=target["device"])
    for t in range(x.shape[0]):
        net2.forward(x[t])
        if t % stride == 0:
            edge2.update(net2["rnn"]["out"], y[t], net2["out"]["out"])
    # (the sweep above may have taken the blocked form -- the default on the symmetric pass with beta = 1 -- which rounds z
    # differently from a pass per update: 1.1e-5 on weights of size 0.6; a pass per update on both sides: < 1e-5)
    assert np.abs(edge2.weights.cpu().numpy() - edge.weights.cpu().numpy()).max() < 2e-5
    assert np.abs(edge2.P.cpu().numpy() - edge.P.cpu().numpy()).max() < 1e-5


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_fit_rls_epochs_recovers_readout(target):
    """mirrors the rls leg of rectipy_tests/test_network.py::test_3_7_optimization"""
    rng = np.random.default_rng(7)
    n, k, steps = 10, 3, 100
    W, W_out, x = rng.standard_normal((n, n)), rng.standard_normal((k, n)), rng.standard_normal((steps, n))
    net = Network(1e-2, **target)
    net.add_diffeq_node("rnn", TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in",
                        input_var="li_op/I_ext", output_var="tanh_op/r", dtype=torch.float32)
    net.add_func_node("output", k, "identity")
    net.add_edge("rnn", "output", weights=W_out)
    y0 = {kk: v.clone() for kk, v in net.state.items()}
    tgt = net.run(x, sampling_steps=1, enable_grad=False, verbose=False).to_numpy("out")
    net.pop_edge("rnn", "output")
    net.add_edge("rnn", "output", train="rls", beta=0.99, alpha=1.0)
    net.reset(y0)
    obs = net.fit_rls([x] * 40, [tgt] * 40, update_steps=1, verbose=False)
    fitted = net.get_edge("rnn", "output").weights.cpu().numpy()
    assert np.mean((W_out - fitted) ** 2) < 0.5 and len(obs["epoch_loss"]) == 40
    with pytest.raises(ValueError):
        net.add_edge("rnn", "output", train="rls", beta=1.5)


@pytest.mark.gpu
@pytest.mark.parametrize("n_in,n_out,T,stride,tiled", [(40, 3, 30, 1, False), (256, 2, 25, 3, False), (1000, 4, 12, 1, False),
                                                       (1030, 1, 9, 2, False), (256, 2, 25, 3, True), (1000, 4, 12, 1, True),
                                                       (4100, 2, 6, 1, True), (9000, 3, 4, 2, None),
                                                       (256, 2, 25, 3, "sym"), (1000, 4, 12, 1, "sym"), (2048, 2, 8, 1, "sym"),
                                                       (4100, 2, 6, 1, "sym"), (3140, 3, 7, 2, "sym"), (9000, 3, 9, 2, None),
                                                       # n_in % 1024 == 0: the unpredicated kernel; 8192 and 9000: enough
                                                       # tiles for the residency cap (one workgroup per CU); "sym-wide":
                                                       # the cap lifted, eight rows in flight
                                                       (4096, 2, 6, 1, "sym"), (8192, 1, 5, 2, "sym"), (4096, 1, 4, 1, "sym-wide")])
def test_rls_sweep_bit_exact(oracle_backend, hip_device, measure_backend, n_in, n_out, T, stride, tiled):
    """Row-per-wave pass, the tiled pass (P walked in address order by one-shot workgroups, 1024-column tiles with
    per-tile partial sums; chosen by size from n_in = 8192) and the symmetric pass (only the blocks on and above the
    diagonal are streamed; row and column partials; mirror at the end: "sym", and by size from 4 updates per call),
    ragged last tiles / chunks / blocks included."""
    gen = torch.Generator().manual_seed(n_in + T)
    X = torch.randn(T, n_in, generator=gen)
    Y = torch.randn(T, n_out, generator=gen)
    W0 = torch.randn(n_out, n_in, generator=gen) * 0.1
    res = []
    knob = {"sym-wide": 8 | (15 << 8)}.get(tiled, 0)
    tiled = "sym" if knob else tiled
    # "sym-wide" runs on the measurement build with its knob set; everything else on the product library
    for backend, dev in ((oracle_backend, "cpu"), (measure_backend if knob else None, hip_device)):
        edge = RLS(n_in, n_out, weights=W0.clone(), beta=0.97, alpha=1.5, backend=backend).to(dev)
        edge.tiled = True if tiled == "sym" else tiled
        edge.symmetric = True if tiled == "sym" else None
        if backend is measure_backend:
            measure_backend.lib.rp_set_rls_sym(knob)
        try:
            pred, loss = edge.sweep(X.to(dev), Y.to(dev), stride)
        finally:
            measure_backend.lib.rp_set_rls_sym(0)
        assert (edge._partials is not None) == (bool(tiled) or (tiled is None and n_in >= RLS.TILED_MIN_N))
        updates = (T + stride - 1) // stride
        assert (edge._colparts is not None) == (tiled == "sym" or (tiled is None and n_in >= RLS.SYMMETRIC_MIN_N
                                                                   and updates >= RLS.SYMMETRIC_MIN_UPDATES))
        if edge._colparts is not None:  # the caller sees a full P again: mirrored between the blocks
            Pm = edge.P.cpu()
            blk = torch.arange(n_in) // 1024
            off = blk[:, None] != blk[None, :]
            assert torch.equal(Pm[off], Pm.T[off])
        res.append([t.cpu().numpy() for t in (pred, loss, edge.weights, edge.P)])
    for a, b, name in zip(res[0], res[1], ("pred", "loss", "weights", "P")):
        same = (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))
        assert same.all(), f"{name}: {np.count_nonzero(~same)} of {a.size} differ, max abs {np.abs(a - b).max()}"


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("n,tiled", [(520, False), (520, True), (2060, False), (2060, True), (1030, False), (777, False),
                                     (2060, "symmetric"), (520, "symmetric"), (2060, "blocked8"), (2060, "blocked4"),
                                     (520, "blocked2")],
                         ids=["n520-rows", "n520-tiled", "n2060-rows", "n2060-tiled", "n1030-4byte", "n777-4byte",
                              "n2060-symmetric", "n520-symmetric", "n2060-blocked8", "n2060-blocked4", "n520-blocked2"])
def test_g31_fit_rls_at_kernel_sizes_matches_reference(target, n, tiled):
    """Golden G31 (tests/golden/make_golden_rls_large.py): the reference's fit_rls at sizes where the 16-byte row pass
    (full 256-column chunks), the tiled pass (two full 1024-column tiles + a ragged one at n = 2060) and the 4-byte pass
    (n % 4 != 0) are the code that runs.  Inputs are regenerated from tests/golden/recipes.py and checked against the
    stored fingerprints first.  Tolerances: a few times the reference's own float32-vs-float64 deviation (printed by
    the generator: W 7e-6 .. 1.1e-5, P 2e-7 relative, predictions 7e-5 .. 1.3e-4); measured here: W <= 1.3e-5, P <= 3.7e-7
    relative, predictions <= 2.9e-5, sum of squares of P to 1e-8."""
    import sys
    sys.path.insert(0, GOLD)
    import recipes
    g = dict(np.load(os.path.join(GOLD, f"g31_rls_n{n}.npz")))
    T, stride, kout = int(g["T"]), int(g["stride"]), 3
    P = recipes.li_network(n, seed=3100 + n, g=0.8, defaults=False)
    x = (0.5 * recipes.normal((T, n), seed=3150 + n)).astype(np.float32)
    tgt = recipes.normal((T, kout), seed=3160 + n)
    W0 = (0.1 * recipes.normal((kout, n), seed=3170 + n)).astype(np.float32)
    for name, arr in (("W", P["W"]), ("x", x), ("target", tgt), ("W0", W0)):
        assert np.array_equal(recipes.fingerprint(arr), g["fp_" + name]), f"regenerated {name} differs from the generator's"
    net, edge = build(dict(dt=g["dt"], W0=W0, beta=g["beta"], alpha=g["alpha"], **P), target)
    edge.symmetric = tiled == "symmetric" or str(tiled).startswith("blocked")
    edge.tiled = bool(tiled)
    # "blockedK": the blocked symmetric sweep -- P streamed once per K updates, the z vectors inside a block from Y = P_b X
    edge.block = int(tiled[len("blocked"):]) if str(tiled).startswith("blocked") else 1
    obs = net.fit_rls(x, tgt, update_steps=stride, sampling_steps=1, verbose=False)
    W_ref, rows = g["W_after_f64"], g["rows"]
    Pm = edge.P.cpu().numpy()
    p_scale = np.abs(g["P_rows_f64"]).max()
    assert np.abs(edge.weights.cpu().numpy() - W_ref).max() < 5e-5 * max(1.0, np.abs(W_ref).max())
    assert np.abs(Pm[rows] - g["P_rows_f64"]).max() < 2e-6 * p_scale
    assert np.abs(np.diag(Pm) - g["P_diag_f64"]).max() < 2e-6 * p_scale
    fp, fp_ref = recipes.fingerprint(Pm), g["P_fp_f64"]
    assert abs(fp[1] - fp_ref[1]) < 1e-6 * fp_ref[1]              # sum of squares of all n^2 entries
    assert np.abs(obs.to_numpy("out") - g["pred_f64"]).max() < 1.5e-4
    assert np.abs(obs.to_numpy("loss") - g["loss_f64"]).max() < 2e-5 * max(1.0, g["loss_f64"].max())
    assert np.abs(net.state["rnn"].cpu().numpy() - g["state_f64"]).max() < 2e-5


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("block", [1, 8], ids=["pass-per-update", "blocked8"])
def test_g34_fit_rls_n8200_default_symmetric_pass_matches_reference(target, block):
    """Golden G34 (tests/golden/make_golden_rls_n8200.py): the reference's fit_rls at n_in = 8200, where this package's
    DEFAULTS select the symmetric pass (n_in >= 8192, >= 4 updates per call) -- blocks on and above the diagonal only,
    row + column partials, the mirror, on the GPU the one-workgroup-per-CU residency cap; 8 full blocks + a ragged one.
    Tolerances as for G31 (float32 engine vs the reference's float64 run)."""
    import sys
    sys.path.insert(0, GOLD)
    import recipes
    g = dict(np.load(os.path.join(GOLD, "g34_rls_n8200.npz")))
    n, T, stride, kout = int(g["n"]), int(g["T"]), int(g["stride"]), 3
    P = recipes.li_network(n, seed=3400 + n, g=0.8, defaults=False)
    x = (0.5 * recipes.normal((T, n), seed=3450 + n)).astype(np.float32)
    tgt = recipes.normal((T, kout), seed=3460 + n)
    W0 = (0.1 * recipes.normal((kout, n), seed=3470 + n)).astype(np.float32)
    for name, arr in (("W", P["W"]), ("x", x), ("target", tgt), ("W0", W0)):
        assert np.array_equal(recipes.fingerprint(arr), g["fp_" + name]), f"regenerated {name} differs from the generator's"
    net, edge = build(dict(dt=g["dt"], W0=W0, beta=g["beta"], alpha=g["alpha"], **P), target)
    edge.block = block   # (12 updates: a full block of 8 and a ragged one of 4)
    obs = net.fit_rls(x, tgt, update_steps=stride, sampling_steps=1, verbose=False)
    assert edge._colparts is not None, "the defaults must have taken the symmetric pass at this size"
    Pm = edge.P.cpu().numpy()
    rows, p_scale = g["rows"], np.abs(g["P_rows_f64"]).max()
    W_ref = g["W_after_f64"]
    assert np.abs(edge.weights.cpu().numpy() - W_ref).max() < 5e-5 * max(1.0, np.abs(W_ref).max())
    assert np.abs(Pm[rows] - g["P_rows_f64"]).max() < 2e-6 * p_scale
    assert np.abs(np.diag(Pm) - g["P_diag_f64"]).max() < 2e-6 * p_scale
    blk = np.arange(n) // 1024
    off = blk[:, None] != blk[None, :]
    assert np.array_equal(Pm[off], Pm.T[off]), "mirrored: the off-diagonal blocks are exact transposes"
    # (inside a diagonal block both halves are updated, as (k z_i) z_j and (k z_j) z_i: symmetric to rounding)
    assert np.abs(Pm - Pm.T).max() < 1e-6 * p_scale
    fp, fp_ref = recipes.fingerprint(Pm), g["P_fp_f64"]
    assert abs(fp[1] - fp_ref[1]) < 1e-6 * fp_ref[1]              # sum of squares of all n^2 entries
    assert np.abs(obs.to_numpy("out") - g["pred_f64"]).max() < 1.5e-4
    assert np.abs(obs.to_numpy("loss") - g["loss_f64"]).max() < 2e-5 * max(1.0, g["loss_f64"].max())
    assert np.abs(net.state["rnn"].cpu().numpy() - g["state_f64"]).max() < 2e-5


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_mirror_is_deferred_until_P_is_read_and_asymmetric_P_keeps_the_dense_pass(target):
    """The symmetric pass leaves the blocks below the diagonal stale (rp_rls.defer_mirror) and `RLS.P` restores them on the
    first read: sweeps back to back -- nothing reads P in between -- give the bits of sweeps with a read in between.  A P
    assigned from outside that is NOT symmetric is not handed to the symmetric pass (its lower blocks would be ignored and
    overwritten): such an edge keeps the dense passes and equals an edge forced onto them."""
    dev, backend = target["device"], target["backend"]
    n, k, T = 2060, 2, 12
    gen = torch.Generator().manual_seed(3)
    X, Y = torch.randn(T, n, generator=gen).to(dev), torch.randn(T, k, generator=gen).to(dev)
    W0 = torch.randn(k, n, generator=gen) * 0.1

    def make(**kw):
        e = RLS(n, k, weights=W0.clone(), beta=0.98, alpha=1.3, backend=backend).to(dev)
        for name, v in kw.items():
            setattr(e, name, v)
        return e

    a, b = make(tiled=True, symmetric=True), make(tiled=True, symmetric=True)
    a.sweep(X[:6], Y[:6], 1)
    assert a._upper_only                      # the mirror has not run ...
    a.sweep(X[6:], Y[6:], 1)
    b.sweep(X[:6], Y[:6], 1)
    Pb = b.P                                  # ... here it has (a read in between)
    assert not b._upper_only and Pb is b.P
    b.sweep(X[6:], Y[6:], 1)
    assert torch.equal(a.P, b.P) and torch.equal(a.weights, b.weights) and not a._upper_only
    blk = torch.arange(n) // 1024
    off = (blk[:, None] != blk[None, :]).to(dev)
    assert torch.equal(a.P[off], a.P.T[off])
    # a dense pass after a symmetric one reads ALL of P: the mirror runs first
    c, d = make(tiled=True, symmetric=True), make(tiled=True, symmetric=True)
    c.sweep(X[:6], Y[:6], 1)
    c.symmetric = False
    c.sweep(X[6:], Y[6:], 1)
    d.sweep(X[:6], Y[:6], 1)
    d.P
    d.symmetric = False
    d.sweep(X[6:], Y[6:], 1)
    assert torch.equal(c.P, d.P)
    # an asymmetric P from outside
    P0 = torch.eye(n) * 1.3
    P0[5, 1500] = 0.25                        # (only one of the two mirror positions)
    e, f = make(tiled=True, symmetric=None), make(tiled=True, symmetric=False)
    e.SYMMETRIC_MIN_N, e.SYMMETRIC_MIN_UPDATES = 0, 1   # by size this edge would take the symmetric pass
    e.P, f.P = P0.clone().to(dev), P0.clone().to(dev)
    e.sweep(X, Y, 1)
    f.sweep(X, Y, 1)
    assert e._P_symmetric is False and not e._upper_only
    assert torch.equal(e.P, f.P) and torch.equal(e.weights, f.weights)
    # ... and a symmetric one from outside is accepted
    g_ = make(tiled=True, symmetric=None)
    g_.SYMMETRIC_MIN_N, g_.SYMMETRIC_MIN_UPDATES = 0, 1
    g_.P = (torch.eye(n) * 1.3).to(dev)
    g_.sweep(X, Y, 1)
    assert g_._P_symmetric is True and g_._upper_only
    # an IN-PLACE edit through the `P` property between sweeps (r05): what was known about P's symmetry is void -- an edit that
    # breaks it sends the next sweep to the dense passes (on every element, as the reference), one that keeps it stays symmetric
    h, i_ = make(tiled=True, symmetric=None), make(tiled=True, symmetric=False)
    h.SYMMETRIC_MIN_N, h.SYMMETRIC_MIN_UPDATES = 0, 1
    h.sweep(X[:6], Y[:6], 1)
    i_.sweep(X[:6], Y[:6], 1)
    assert h._upper_only and h._P_symmetric is True
    i_.P = h.P.clone()                        # (the same start for the comparison: all of P, mirrored)
    i_.weights.copy_(h.weights)
    h.P[7, 1800] += 0.5                       # one of the two mirror positions
    i_.P[7, 1800] += 0.5
    h.sweep(X[6:], Y[6:], 1)
    i_.sweep(X[6:], Y[6:], 1)
    assert h._P_symmetric is False and not h._upper_only
    assert torch.equal(h.P, i_.P) and torch.equal(h.weights, i_.weights)
    j_ = make(tiled=True, symmetric=None)
    j_.SYMMETRIC_MIN_N, j_.SYMMETRIC_MIN_UPDATES = 0, 1
    j_.sweep(X[:6], Y[:6], 1)
    j_.P.mul_(1.01)                           # keeps the symmetry
    j_.sweep(X[6:], Y[6:], 1)
    assert j_._P_symmetric is True and j_._upper_only


@pytest.mark.gpu
@pytest.mark.parametrize("n_in,n_out,T,stride,block,rows", [
    (256, 2, 25, 3, 2, 64), (1000, 4, 12, 1, 4, 128), (2048, 2, 19, 1, 8, 64), (4100, 2, 11, 1, 4, 64), (3140, 3, 21, 2, 8, 128),
    (4096, 2, 17, 1, 8, 64), (8192, 1, 18, 2, 8, 128), (9000, 3, 20, 1, 8, 128), (9000, 2, 9, 1, 2, 64), (9000, 2, 9, 1, 4, 64),
    (3140, 2, 19, 1, 8, 256), (4096, 3, 10, 2, 4, 256), (2300, 2, 9, 1, 2, 256)])
def test_blocked_rls_sweep_bit_exact(oracle_backend, hip_device, n_in, n_out, T, stride, block, rows):
    """The blocked symmetric sweep (rp_rls.block: P streamed once per `block` updates, the block before's rank-1 updates and
    the products with all inputs of the block in one pass; the z vectors of the block by the double-precision recurrence)
    on the GPU against the C oracle, bit for bit: full and ragged blocks of updates, ragged tiles / chunks / 1024-blocks, 64 / 128 / 256 rows per tile, the
    unpredicated kernel (n_in % 1024 == 0), enough tiles for the residency cap (8192, 9000), two consecutive sweeps."""
    gen = torch.Generator().manual_seed(n_in + T)
    base = torch.randn(n_in, generator=gen)
    X = base[None, :] + 0.1 * torch.randn(T, n_in, generator=gen)   # consecutive inputs alike: z_j is a small difference
    Y = torch.randn(T, n_out, generator=gen)
    W0 = torch.randn(n_out, n_in, generator=gen) * 0.1
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        edge = RLS(n_in, n_out, weights=W0.clone(), beta=0.97, alpha=1.5, backend=backend).to(dev)
        edge.tiled, edge.symmetric, edge.block = True, True, block
        edge.BLOCK_ROWS = rows
        half = (T // 2) // stride * stride
        p1, l1 = edge.sweep(X[:half].to(dev), Y[:half].to(dev), stride)
        p2, l2 = edge.sweep(X[half:].to(dev), Y[half:].to(dev), stride)
        assert edge._blk is not None and edge._upper_only
        res.append([t.cpu().numpy() for t in (torch.cat([p1, p2]), torch.cat([l1, l2]), edge.weights, edge.P)])
    for a, b, name in zip(res[0], res[1], ("pred", "loss", "weights", "P")):
        same = (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))
        assert same.all(), f"{name}: {np.count_nonzero(~same)} of {a.size} differ, max abs {np.abs(a - b).max()}"


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_blocked_sweep_is_the_same_recursion_and_no_less_accurate(target):
    """37 updates with inputs that differ little from step to step (what a slowly moving population feeds a read-out: z is
    then a small difference of large vectors), against the recursion in float64 (reference edges.py:227-234): the blocked
    sweep -- its z recurrence runs in double precision -- deviates no more than one pass per update does (measured: predictions
    4.3e-5 against 3.5e-4, weights 4.4e-7 against 8.2e-7, P 6.8e-7 both)."""
    n_in, n_out, T = 1500, 3, 37
    gen = torch.Generator().manual_seed(5)
    base = torch.randn(n_in, generator=gen) * 0.5
    X = base[None, :] + 0.05 * torch.randn(T, n_in, generator=gen)
    Y = torch.randn(T, n_out, generator=gen)
    P, W, preds = 1.5 * np.eye(n_in), np.zeros((n_out, n_in)), []
    Xd, Yd = X.double().numpy(), Y.double().numpy()
    for t in range(T):
        x = Xd[t]
        z = (1 / 0.98) * (P @ x)
        k = 1 / (1 + x @ z)
        yh = W @ x
        preds.append(yh)
        W = W + np.outer(k * (Yd[t] - yh), z)
        P = P - k * np.outer(z, z)
    dev = {}
    for block in (1, 8):
        edge = RLS(n_in, n_out, beta=0.98, alpha=1.5, backend=target["backend"]).to(target["device"])
        edge.tiled, edge.symmetric, edge.block = True, True, block
        pred, loss = edge.sweep(X.to(target["device"]), Y.to(target["device"]), 1)
        dev[block] = (np.abs(pred.cpu().numpy() - np.array(preds)).max(), np.abs(edge.weights.cpu().numpy() - W).max(),
                      np.abs(edge.P.cpu().numpy() - P).max())
    assert dev[8][0] < 1e-4 and dev[8][1] < 2e-6 and dev[8][2] < 2e-6, dev
    assert all(b <= 1.5 * a + 1e-7 for a, b in zip(dev[1], dev[8])), dev


def test_default_form_follows_beta(oracle_backend):
    """On the symmetric pass the blocked sweep is the default for the reference's default beta = 1; with beta < 1 (a recursion
    that amplifies rounding in any form) the literal pass per update stays the default and the blocked sweep is a choice."""
    gen = torch.Generator().manual_seed(0)
    X, Y = torch.randn(12, 64, generator=gen), torch.randn(12, 2, generator=gen)
    used = {}
    for beta, block in ((1.0, None), (0.97, None), (0.97, 8)):
        edge = RLS(64, 2, beta=beta, alpha=1.0, backend=oracle_backend)
        edge.tiled, edge.symmetric, edge.block = True, True, block
        edge.sweep(X, Y, 1)
        used[(beta, block)] = edge._blk is not None
    assert used == {(1.0, None): True, (0.97, None): False, (0.97, 8): True}
