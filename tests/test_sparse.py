"""Packed storage of a fixed sparse coupling matrix (rectipy_amd/sparse.py): the layout is checked on the CPU against a
scalar emulation of the sparse kernel's arithmetic and the oracle's dense row dot (bit for bit); on the GPU the sparse
step kernel must reproduce the dense oracle exactly for every model family, with the LDS and the L2 gather paths."""
import numpy as np
import pytest
import torch

from rectipy_amd import _abi, sparse
from rectipy_amd.engine import Population


def random_sparse(rows, n, p, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(rows, n, generator=g) * scale
    return W * (torch.rand(rows, n, generator=g) < p)


def emulate_row(packed, src, i):
    """What one wave of step_sparse_kernel computes for row i: every lane walks its entry list (4 per group) and feeds
    the accumulator of the entry's component; then (a0+a1)+(a2+a3) per lane and the 64-lane butterfly."""
    vals, code, off = packed.vals.numpy(), packed.code.numpy().view(np.uint16), packed.rowoff.numpy()
    a = np.zeros((64, 4), dtype=np.float32)
    for k in range(off[i], off[i + 1]):
        for l in range(64):
            for e in range(4):
                v, w = vals[k * 256 + 4 * l + e], int(code[k * 256 + 4 * l + e])
                c, m = w >> 14, w & 0x3fff
                x = src[256 * m + 4 * l + c]
                a[l, c] = np.float32(np.float64(v) * np.float64(x) + np.float64(a[l, c]))  # fmaf: one rounding
    s = np.array([np.float32(np.float32(a[l, 0] + a[l, 1]) + np.float32(a[l, 2] + a[l, 3])) for l in range(64)],
                 dtype=np.float32)
    for offb in (32, 16, 8, 4, 2, 1):
        s = np.array([np.float32(s[l] + s[l ^ offb]) for l in range(64)], dtype=np.float32)
    return s[0]


@pytest.mark.parametrize("rows,n,row0,p", [(24, 256, 0, 0.1), (40, 1028, 8, 0.2), (16, 2052, 64, 0.05), (8, 512, 0, 1.0)])
def test_packed_rows_reproduce_the_dense_row_dot(oracle_backend, rows, n, row0, p):
    W = random_sparse(rows, n, p, seed=rows + n)
    W[3] = 0.0                                   # an empty row
    packed = sparse.pack_rows(W, row0=row0, block_rows=16)
    off = packed.rowoff.numpy()
    assert off[0] == 0 and off[-1] * 256 == packed.vals.numel() and (np.diff(off) >= 0).all()
    assert off[4] == off[3]                      # the empty row stores nothing
    assert int((packed.vals != 0).sum()) == int((W != 0).sum())
    src = torch.randn(n + 256, generator=torch.Generator().manual_seed(1)).numpy().astype(np.float32)
    import ctypes as C
    lib = oracle_backend.lib
    for i in range(rows):
        w = np.ascontiguousarray(W[i].numpy())
        dense = lib.oracle_row_dot(w.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), n, 1, row0 + i)
        got = emulate_row(packed, src, i)
        assert np.float32(dense).view(np.uint32) == np.float32(got).view(np.uint32) or (dense == 0 and got == 0), i


def emulate_block(packed, src, b, rows):
    """What one wave of the blocked kernel computes for block b: lanes walk the block's stream at their own pace, park
    the lane sum at every END entry; afterwards one butterfly per row."""
    vals, code, off = packed.vals.numpy(), packed.code.numpy().view(np.uint16), packed.rowoff.numpy()
    RB = packed.block_rows
    nr = min(RB, rows - b * RB)
    stash = np.zeros((RB + 1, 64), dtype=np.float32)
    for l in range(64):
        a, r = np.zeros(4, dtype=np.float32), 0
        for k in range(off[b], off[b + 1]):
            for e in range(4):
                v, w = vals[k * 256 + 4 * l + e], int(code[k * 256 + 4 * l + e])
                c, m = w >> 14, w & 0x1fff
                a[c] = np.float32(np.float64(v) * np.float64(src[256 * m + 4 * l + c]) + np.float64(a[c]))
                if w & sparse.END_FLAG:
                    stash[r, l] = np.float32(np.float32(a[0] + a[1]) + np.float32(a[2] + a[3]))
                    a[:] = 0
                    r += 1
        assert r == nr, f"lane {l} of block {b} closed {r} row segments, the block has {nr} rows"
    out = []
    for r in range(nr):
        s = stash[r].copy()
        for offb in (32, 16, 8, 4, 2, 1):
            s = np.array([np.float32(s[l] + s[l ^ offb]) for l in range(64)], dtype=np.float32)
        out.append(s[0])
    return out


@pytest.mark.parametrize("rows,n,row0,p,rb", [(24, 256, 0, 0.1, 16), (40, 1028, 8, 0.2, 16), (21, 2052, 64, 0.02, 4),
                                              (8, 512, 0, 1.0, 8)])
def test_blocked_packed_rows_reproduce_the_dense_row_dot(oracle_backend, rows, n, row0, p, rb):
    """Blocked form: row blocks share one padded stream, END entries close a lane's row segments (dummy END entries for
    empty segments, an all-zero row included)."""
    W = random_sparse(rows, n, p, seed=rows + n)
    W[3] = 0.0
    packed = sparse.pack_rows_blocked(W, row0=row0, block_rows=rb, proc_rows=2 * rb)
    off = packed.rowoff.numpy()
    nb = -(-rows // rb)
    assert packed.block_rows == rb and off.size == nb + 1 and off[-1] * 256 == packed.vals.numel()
    assert int((packed.vals != 0).sum()) == int((W != 0).sum())
    ends = (packed.code.numpy().view(np.uint16) & sparse.END_FLAG) != 0
    assert int(ends.sum()) == rows * 64          # every (row, lane) segment is closed exactly once
    src = torch.randn(n + 256, generator=torch.Generator().manual_seed(1)).numpy().astype(np.float32)
    import ctypes as C
    lib = oracle_backend.lib
    for b in range(nb):
        got = emulate_block(packed, src, b, rows)
        for j, g in enumerate(got):
            i = b * rb + j
            w = np.ascontiguousarray(W[i].numpy())
            dense = lib.oracle_row_dot(w.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), n, 1, row0 + i)
            assert np.float32(dense).view(np.uint32) == np.float32(g).view(np.uint32) or (dense == 0 and g == 0), i


def test_blocked_form_pads_less_than_the_per_row_form():
    W = random_sparse(128, 32768, 0.1, seed=9)
    ideal = 6 * int((W != 0).sum())
    per_row, b8, b16 = sparse.pack_rows(W), sparse.pack_rows_blocked(W, block_rows=8, proc_rows=64), \
        sparse.pack_rows_blocked(W, block_rows=16, proc_rows=64)
    assert per_row.stored_bytes > 1.3 * ideal and b8.stored_bytes < 1.13 * ideal and b16.stored_bytes < 1.09 * ideal


def test_pack_rejects_unaligned_width():
    with pytest.raises(ValueError):
        sparse.pack_rows(torch.zeros(4, 10))


MODELS = {
    "li_tanh": (dict(tau=8.0, k=1.0, eta=0.1), dict(), 1e-2, 0.5, _abi.RP_OUT_SOURCE),
    "qif": (dict(tau=1.0, k=1.0, eta=-5.0, tau_s=1.0), dict(v0=-2.0, s0=0.0, theta=20.0, v_reset=-20.0), 1e-2, 15.0,
            _abi.RP_OUT_SOURCE),
    "ik": (dict(C=100.0, k=0.7, v_r=-60.0, v_theta=-40.0, eta=0.0, g=1.0, E_r=0.0, b=-2.0, tau_u=33.33, kappa=10.0,
                tau_s=6.0), dict(v0=-60.0, s0=0.0, aux0=0.0, theta=40.0, v_reset=-55.0), 1e-1, 120.0, _abi.RP_OUT_STATE0),
}


@pytest.mark.gpu
@pytest.mark.parametrize("model", list(MODELS))
@pytest.mark.parametrize("rows,n,row0,form", [(256, 256, 0, "blocked8"), (1028, 1028, 0, "blocked16"), (1000, 4100, 0, "blocked4"),
                                              (256, 256, 0, "per_row"), (1028, 1028, 0, "per_row"),
                                              (96, 33024, 64, "windowed")])
def test_sparse_step_kernel_equals_dense_oracle(oracle_backend, hip_device, model, rows, n, row0, form):
    """n <= 32768: source vector gathered from LDS (rotated layout), blocked form (row blocks share one padded stream) or
    the per-row form; the 96 x 33024 shard takes the windowed form."""
    params, kw, dt, drive, out_var = MODELS[model]
    T = 600 if model == "ik" else 200
    positive = model != "li_tanh"
    W = random_sparse(rows, n, 0.1, seed=n + rows, scale=(20.0 / n if positive else 1.5 / (0.1 * n) ** 0.5))
    if positive:
        W = W.abs()
    x = torch.full((T, 1), drive) + torch.randn(T, 1, generator=torch.Generator().manual_seed(2)) * 0.1 * abs(drive)
    src0 = torch.rand(n, generator=torch.Generator().manual_seed(3)) * 0.5
    res = []
    for backend, dev, pack in ((oracle_backend, "cpu", False), (None, hip_device, True)):
        pop = Population(model, W, params, dt, dev, n=n, row0=row0, backend=backend, **kw)
        if pack:
            pop.blocked_sparse = form.startswith("blocked")
            pop.sparse_block_rows = int(form[7:]) if pop.blocked_sparse else 0
            packed = pop.pack_weights()
            assert packed.block_rows == pop.sparse_block_rows and (packed.row_ids is not None) == (form == "windowed")
            if n >= 8192:  # the slot layout needs many chunks per row (n / 256 >> 1 / density) to pay off
                assert packed.stored_bytes < 0.4 * 4 * rows * n
        traj = torch.zeros(T, rows, device=dev)
        if rows == n:
            pop.forward(T, x=x.to(dev), traj=traj, out_var=out_var)
        else:  # a row shard: the rest of the source vector comes from outside every step
            for t in range(T):
                own = pop.source_local.clone()
                pop.source.copy_(src0.to(dev) * (1 + 0.01 * t))
                pop.set_source_local(own)
                pop.forward(1, x=x[t:t + 1].to(dev), traj=traj[t:t + 1], out_var=out_var, step0=t)
        if dev != "cpu":
            torch.cuda.synchronize()
        res.append((traj.cpu(), pop.v.cpu(), pop.source_local.cpu()))
    for a, b, name in zip(res[0], res[1], ("trajectory", "v", "source")):
        a, b = a.numpy(), b.numpy()
        same = (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))
        assert same.all(), f"{model} {rows}x{n} {name}: {np.count_nonzero(~same)} differ, max {np.abs(a - b).max()}"
    assert np.isfinite(res[0][0].numpy()).all() and np.abs(np.diff(res[0][0].numpy(), axis=0)).max() > 0, "no dynamics"
    if model != "li_tanh":  # the spiking runs must actually spike: the synaptic variable s has been kicked
        assert res[0][2].numpy().max() > 0.05, "no spikes in the window"


@pytest.mark.gpu
def test_front_end_packs_a_fixed_sparse_matrix_automatically(hip_device):
    """Network.add_diffeq_node with a p = 0.1 matrix (the reference's random_connectivity use): packed automatically
    when W is not trained, never when it is; outputs identical to the dense path, bit for bit."""
    from rectipy_amd import Network
    N, T = 4096, 300
    W = random_sparse(N, N, 0.1, seed=4, scale=1.5 / (0.1 * N) ** 0.5).numpy()
    x = torch.randn(T, 1, generator=torch.Generator().manual_seed(5)).numpy() * 0.5
    tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
    outs = {}
    for mode, train in (("auto", None), (False, None), ("auto", ["weights"])):
        net = Network(1e-2, device="cuda:0")
        node = net.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="tanh_op/r", weights=W,
                                   source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32,
                                   train_params=train, sparse_weights=mode)
        assert (node._pop.packed is not None) == (mode == "auto" and train is None)
        outs[(mode, bool(train))] = net.run(x, sampling_steps=10, verbose=False, enable_grad=False).to_numpy("out")
    a, b, c = outs[("auto", False)], outs[(False, False)], outs[("auto", True)]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(b.view(np.uint32), c.view(np.uint32))
    assert np.abs(a).max() > 0.01
    # a dense matrix is left alone
    net = Network(1e-2, device="cuda:0")
    node = net.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="tanh_op/r",
                               weights=np.random.default_rng(0).standard_normal((N, N)).astype(np.float32) / N ** 0.5,
                               source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32)
    assert node._pop.packed is None


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["set_var", "in_place"])
def test_weight_writes_reach_the_packed_form(hip_device, how):
    """A packed (auto) node whose weights are replaced through Network.set_var / edited through the live
    ``node["weights"]`` tensor integrates with the NEW matrix, forward and adjoint (the packed copies are rebuilt)."""
    from rectipy_amd import Network
    N, T = 4096, 50
    tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
    W0 = random_sparse(N, N, 0.1, seed=6, scale=1.5 / (0.1 * N) ** 0.5)
    W1 = random_sparse(N, N, 0.1, seed=7, scale=1.5 / (0.1 * N) ** 0.5)
    x = torch.randn(T, 1, generator=torch.Generator().manual_seed(8)).numpy() * 0.5

    def build(W, mode):
        net = Network(1e-2, device="cuda:0")
        node = net.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="tanh_op/r", weights=W.numpy(),
                                   source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32,
                                   sparse_weights=mode, train_params=["li_op/eta"])
        return net, node

    net, node = build(W0, "auto")
    assert node._pop.packed is not None
    net.run(x, sampling_steps=10, verbose=False, enable_grad=False)  # the packed forms are in use
    net.reset()
    if how == "set_var":
        net.set_var("rnn", "weights", W1)
    else:
        with torch.no_grad():
            node["weights"].copy_(W1.to("cuda:0"))
    ref_net, ref_node = build(W1, False)
    res = []
    for nt, nd in ((net, node), (ref_net, ref_node)):
        xs = torch.tensor(x, device="cuda:0")
        traj = nd._grad_window(T, xs)
        traj.square().sum().backward()
        res.append((traj.detach().cpu().numpy(), nd["li_op/eta"].grad.cpu().numpy()))
    assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32)), "forward used a stale packed W"
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-5 * np.abs(res[1][1]).max(), "adjoint used a stale packed W^T"


@pytest.mark.gpu
@pytest.mark.parametrize("rb", [0, 8], ids=["per_row", "blocked8"])
@pytest.mark.parametrize("model,n", [("li_tanh", 1028), ("li_tanh", 8192), ("qif", 1024), ("ik", 2048)])
def test_sparse_adjoint_equals_oracle(oracle_backend, hip_device, model, n, rb):
    """Adjoint sweep on the packed rows of W^T (fixed sparse W; per-row and blocked form): HIP == oracle (row-dot order
    on W^T), bit for bit, for the rate and the spiking sweeps; n = 8192 is beyond the dense transposed-copy threshold."""
    params, kw, dt, drive, out_var = MODELS[model]
    T = 12 if n > 4096 else 40
    positive = model != "li_tanh"
    W = random_sparse(n, n, 0.1, seed=n + 1, scale=(20.0 / n if positive else 1.5 / (0.1 * n) ** 0.5))
    if positive:
        W = W.abs()
    gen = torch.Generator().manual_seed(6)
    x = torch.full((T, 1), drive) + torch.randn(T, 1, generator=gen) * 0.1 * abs(drive)
    g_out = torch.randn(T, n, generator=gen) * 0.1
    l0 = [torch.randn(n, generator=gen) * 0.01 for _ in range(3)]
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        pop = Population(model, W, params, dt, dev, backend=backend, **kw)
        pop.blocked_sparse, pop.sparse_block_rows = rb > 0, rb
        pop.pack_weights()
        tv, ts, ta, tx, ut, gx = (torch.zeros(T, n, device=dev) for _ in range(6))
        pop.forward(T, x=x.to(dev), save_v=tv, save_src=ts, save_acc=ta, save_aux=tx if pop.aux is not None else None)
        lv, ls, lx = (t_.clone().to(dev) for t_ in l0)
        if model == "li_tanh":
            gk = torch.zeros(n, device=dev)
            keep = pop.backward(tv, ts, g_out.to(dev), _abi.RP_OUT_SOURCE, lv, ut, traj_acc=ta, grad_k=gk, grad_x=gx)
            out = [lv, ut, gk, gx]
        else:
            grads = {"eta": torch.zeros(n, device=dev), "k": torch.zeros(n, device=dev)}
            keep = pop.backward(tv, ts, g_out.to(dev), _abi.RP_OUT_SOURCE, lv, ut, traj_acc=ta, grad_x=gx, lam_s=ls,
                                slope=0.7, x=x.to(dev), traj_aux=tx if pop.aux is not None else None,
                                lam_x=lx if pop.aux is not None else None, grad_params=grads)
            out = [lv, ls, lx, ut, gx, grads["eta"], grads["k"]]
        assert pop.packed_t is not None and pop.packed_t.block_rows == rb and pop.packed.block_rows == rb
        if dev != "cpu":
            torch.cuda.synchronize()
        res.append([t_.cpu().numpy() for t_ in out])
        del keep
    for i, (a, b) in enumerate(zip(*res)):
        same = (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))
        assert same.all(), f"{model} n={n} output {i}: {np.count_nonzero(~same)} differ, max {np.abs(a - b).max()}"
    assert np.abs(res[0][0]).max() > 0


@pytest.mark.gpu
def test_sparse_and_dense_adjoints_agree(hip_device):
    """Below the transposed-copy threshold both layouts use the row-dot order on W^T: identical bits.  Above it the
    dense sweep sums column strips: equal to rounding."""
    for n, exact in ((2048, True), (8192, False)):
        W = random_sparse(n, n, 0.1, seed=n, scale=1.5 / (0.1 * n) ** 0.5)
        T = 10
        x = torch.randn(T, 1, generator=torch.Generator().manual_seed(1)) * 0.5
        g_out = torch.randn(T, n, generator=torch.Generator().manual_seed(2)) * 0.1
        outs = []
        for sparse_adj in (True, False):
            pop = Population("li_tanh", W, dict(tau=8.0, k=1.0, eta=0.1), 1e-2, hip_device)
            pop.pack_weights()
            pop.sparse_adjoint = sparse_adj
            tv, ts, ut = (torch.zeros(T, n, device=hip_device) for _ in range(3))
            pop.forward(T, x=x.to(hip_device), save_v=tv, save_src=ts)
            lam = torch.zeros(n, device=hip_device)
            keep = pop.backward(tv, ts, g_out.to(hip_device), _abi.RP_OUT_SOURCE, lam, ut)
            torch.cuda.synchronize()
            outs.append(lam.cpu().numpy())
            del keep
        if exact:
            assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
        else:
            assert np.abs(outs[0] - outs[1]).max() <= 2e-5 * np.abs(outs[1]).max()


@pytest.mark.gpu
def test_fit_bptt_through_a_fixed_sparse_reservoir(hip_device):
    """Training the input and read-out edges by BPTT through a fixed sparse recurrent matrix (the pattern of the
    reference's documentation/bptt_spiking_neurons_ffwd.py, where the fixed matrix is all zeros): forward and adjoint both
    run on the packed forms; same trained weights as with dense storage."""
    from rectipy_amd import Network
    N, T, m, k = 4096, 120, 3, 2
    g = torch.Generator().manual_seed(8)
    W = random_sparse(N, N, 0.1, seed=9, scale=1.2 / (0.1 * N) ** 0.5).numpy()
    W_in = (torch.randn(N, m, generator=g) * 0.5).numpy()
    W_out = (torch.randn(k, N, generator=g) / N ** 0.5).numpy()
    x = torch.randn(T, m, generator=g).numpy()
    tgt = (torch.randn(T, k, generator=g) * 0.1).numpy()
    tanh = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
    res = []
    for mode in ("auto", False):
        net = Network(1e-2, device="cuda:0")
        node = net.add_diffeq_node("rnn", tanh, input_var="li_op/I_ext", output_var="tanh_op/r", weights=W,
                                   source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32,
                                   sparse_weights=mode)
        net.add_func_node("inp", m, "identity")
        net.add_func_node("out", k, "identity")
        e_in = net.add_edge("inp", "rnn", weights=W_in.copy(), train="gd")
        e_out = net.add_edge("rnn", "out", weights=W_out.copy(), train="gd")
        net.fit_bptt(x, tgt, optimizer="sgd", lr=0.1, update_steps=40, sampling_steps=1, verbose=False)
        torch.cuda.synchronize()
        assert (node._pop.packed is not None) == (mode == "auto")
        assert (node._pop.packed_t is not None) == (mode == "auto")
        res.append((e_in.weights.detach().cpu().numpy(), e_out.weights.detach().cpu().numpy()))
    assert np.abs(res[0][0] - W_in).max() > 1e-6, "the input edge was not trained"
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def emulate_row_windowed(packed, src, n, slot):
    """One row of the windowed sparse kernel: phases in order, each gathering from the window it stages (the code holds
    the chunk index relative to that window)."""
    vals, code, off = packed.vals.numpy(), packed.code.numpy().view(np.uint16), packed.rowoff.numpy()
    NP, Wc = packed.phases, sparse.WINDOW_CHUNKS
    cls = int(packed.batch_class[slot // sparse.BATCH_ROWS])
    sched = packed.phase_window.numpy().reshape(-1, NP)[cls]
    a = np.zeros((64, 4), dtype=np.float32)
    padded = np.zeros(((n + 255) // 256 + Wc) * 256, dtype=np.float32)
    padded[:n] = src[:n]
    for ph in range(NP):
        w = int(sched[ph])
        g0, g1 = off[slot * NP + ph], off[slot * NP + ph + 1]
        assert w >= 0 or g0 == g1
        window = padded[w * Wc * 256:(w + 1) * Wc * 256] if w >= 0 else None
        for k in range(g0, g1):
            for l in range(64):
                for e in range(4):
                    v, word = vals[k * 256 + 4 * l + e], int(code[k * 256 + 4 * l + e])
                    c, m = word >> 14, word & 0x3fff
                    a[l, c] = np.float32(np.float64(v) * np.float64(window[256 * m + 4 * l + c]) + np.float64(a[l, c]))
    s = np.array([np.float32(np.float32(a[l, 0] + a[l, 1]) + np.float32(a[l, 2] + a[l, 3])) for l in range(64)],
                 dtype=np.float32)
    for offb in (32, 16, 8, 4, 2, 1):
        s = np.array([np.float32(s[l] + s[l ^ offb]) for l in range(64)], dtype=np.float32)
    return s[0]


@pytest.mark.parametrize("rows,n,row0", [(24, 256 * 300 + 8, 8), (16, 256 * 256, 1024), (12, 256 * 130, 40)])
def test_windowed_packed_rows_reproduce_the_dense_row_dot(oracle_backend, rows, n, row0):
    """n > 32768: phases / windows / batches of the windowed layout against the oracle's dense row dot, bit for bit
    (three windows + partial chunk; exactly two windows, no partial chunk; rows starting in different windows)."""
    import ctypes as C
    W = random_sparse(rows, n, 0.02, seed=rows + n)
    packed = sparse.pack_rows_windowed(W, row0=row0)
    assert int((packed.vals != 0).sum()) == int((W != 0).sum())
    ids = packed.row_ids.numpy()
    assert sorted(i for i in ids if i >= 0) == list(range(rows)) and len(ids) % sparse.BATCH_ROWS == 0
    src = torch.randn(n, generator=torch.Generator().manual_seed(1)).numpy().astype(np.float32)
    lib = oracle_backend.lib
    for slot, i in enumerate(ids):
        if i < 0:
            assert packed.rowoff[slot * packed.phases] == packed.rowoff[(slot + 1) * packed.phases]
            continue
        w = np.ascontiguousarray(W[i].numpy())
        dense = lib.oracle_row_dot(w.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), n, 1, row0 + int(i))
        got = emulate_row_windowed(packed, src, n, slot)
        assert np.float32(dense).view(np.uint32) == np.float32(got).view(np.uint32) or (dense == 0 and got == 0), i


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["li_tanh", "qif"])
@pytest.mark.parametrize("rows,n,row0", [(200, 33024, 64), (136, 65536, 1024), (96, 256 * 300 + 8, 8)])
def test_windowed_sparse_step_kernel_equals_dense_oracle(oracle_backend, hip_device, model, rows, n, row0):
    """n > 32768 on the windowed form (source vector staged in LDS window by window, rows batched by start window):
    HIP == dense oracle, bit for bit, incl. padding slots, a start exactly on a window edge and the partial chunk."""
    params, kw, dt, drive, out_var = MODELS[model]
    T = 40 if model == "li_tanh" else 160  # the QIF rows need ~60 steps to their first spikes
    positive = model != "li_tanh"
    W = random_sparse(rows, n, 0.05, seed=n + rows, scale=(40.0 / n if positive else 1.5 / (0.05 * n) ** 0.5))
    if positive:
        W = W.abs()
    x = torch.full((T, 1), drive) + torch.randn(T, 1, generator=torch.Generator().manual_seed(2)) * 0.1 * abs(drive)
    src0 = torch.rand(n, generator=torch.Generator().manual_seed(3)) * 0.5
    res = []
    for backend, dev, pack in ((oracle_backend, "cpu", False), (None, hip_device, True)):
        pop = Population(model, W, params, dt, dev, n=n, row0=row0, backend=backend, **kw)
        if pack:
            packed = pop.pack_weights()
            assert packed.row_ids is not None and packed.phases >= 3
        traj = torch.zeros(T, rows, device=dev)
        for t in range(T):  # a row shard: the rest of the source vector comes from outside every step
            own = pop.source_local.clone()
            pop.source.copy_(src0.to(dev) * (1 + 0.01 * t))
            pop.set_source_local(own)
            pop.forward(1, x=x[t:t + 1].to(dev), traj=traj[t:t + 1], out_var=out_var, step0=t)
        if dev != "cpu":
            torch.cuda.synchronize()
        res.append((traj.cpu(), pop.v.cpu(), pop.source_local.cpu()))
    for a, b, name in zip(res[0], res[1], ("trajectory", "v", "source")):
        a, b = a.numpy(), b.numpy()
        same = (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))
        assert same.all(), f"{model} {rows}x{n} {name}: {np.count_nonzero(~same)} differ, max {np.abs(a - b).max()}"
    assert np.isfinite(res[0][0].numpy()).all() and np.abs(np.diff(res[0][0].numpy(), axis=0)).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n", [33024, 65536])
def test_windowed_sparse_adjoint_equals_l2_gather_form(hip_device, n):
    """Adjoint on the packed rows of W^T for n > 32768: the windowed form (LDS staging per window) must give the bits of
    the plain packed form (L2 gather), which tests at smaller n pin against the oracle."""
    T = 4
    gen = torch.Generator(device=hip_device).manual_seed(n)
    W = torch.randn(n, n, device=hip_device, generator=gen) * (1.5 / (0.02 * n) ** 0.5)
    W *= torch.rand(n, n, device=hip_device, generator=gen) < 0.02
    x = torch.randn(T, 1, generator=torch.Generator().manual_seed(1)).to(hip_device) * 0.5
    g_out = torch.randn(T, n, device=hip_device, generator=gen) * 0.1
    outs = []
    for windowed in (True, False):
        pop = Population("li_tanh", W, dict(tau=8.0, k=1.0, eta=0.1), 1e-2, hip_device)
        pop.windowed_sparse = windowed
        pop.pack_weights()
        tv, ts, ta, ut, gx = (torch.zeros(T, n, device=hip_device) for _ in range(5))
        pop.forward(T, x=x, save_v=tv, save_src=ts, save_acc=ta)
        lam, gk = torch.zeros(n, device=hip_device), torch.zeros(n, device=hip_device)
        keep = pop.backward(tv, ts, g_out, _abi.RP_OUT_SOURCE, lam, ut, traj_acc=ta, grad_k=gk, grad_x=gx)
        torch.cuda.synchronize()
        assert (pop.packed_t.row_ids is not None) == windowed
        outs.append([t_.cpu().numpy() for t_ in (tv, lam, ut, gk, gx)])
        del keep, pop
    for a, b in zip(*outs):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.abs(outs[0][1]).max() > 0
