"""B independent trajectories through one coupling matrix (rp_forward_batched, engine.BatchedPopulation): W is streamed
once per step for all of them.  Trajectory b of a batched run must reproduce an UNBATCHED run on the same state bit for
bit (same accumulators, same chunk order, same butterfly), on the oracle (host logic) and on the GPU (kernel)."""
import numpy as np
import pytest
import torch

from rectipy_amd import _abi
from rectipy_amd.engine import BatchedPopulation, Population

TARGETS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture
def target(request, oracle_backend):
    if request.param == "oracle":
        return dict(device="cpu", backend=oracle_backend)
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return dict(device="cuda:0", backend=None)


def network(n, model, seed):
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(n, n, generator=g) * 1.5 / n ** 0.5
    params = dict(tau=torch.rand(n, generator=g) * 5 + 5, k=torch.rand(n, generator=g) + 0.5, eta=torch.randn(n, generator=g) * 0.1)
    if model == "li_sigmoid":
        params.update(r_max=torch.rand(n, generator=g) + 0.5, s=torch.rand(n, generator=g) * 2 + 0.5,
                      v0=torch.randn(n, generator=g) * 0.2)
    return W, params


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("model,n,B,xw", [("li_tanh", 64, 2, "full"), ("li_tanh", 260, 4, "bcast"), ("li_sigmoid", 516, 8, "full"),
                                         ("li_tanh", 1028, 16, "none"), ("li_tanh", 2048, 8, "full")])
def test_batched_equals_unbatched_runs(target, model, n, B, xw):
    dev, backend = target["device"], target["backend"]
    W, params = network(n, model, seed=n + B)
    g = torch.Generator().manual_seed(7)
    T = 25
    v0 = torch.randn(B, n, generator=g)
    x = {"full": torch.randn(T, B, n, generator=g) * 0.5, "bcast": torch.randn(T, B, 1, generator=g), "none": None}[xw]
    pop = Population(model, W, params, 1e-2, dev, backend=backend)
    bp = BatchedPopulation(pop, B, mfma=False)
    bp.v.copy_(v0.to(dev))
    bp.refresh()
    traj = torch.zeros(T, B, n, device=dev)
    bp.forward(10, x=None if x is None else x[:10].to(dev), traj=traj[:10], out_var=_abi.RP_OUT_SOURCE)
    bp.forward(T - 10, x=None if x is None else x[10:].to(dev), traj=traj[10:], out_var=_abi.RP_OUT_SOURCE)
    for b in range(B):
        single = Population(model, W, params, 1e-2, dev, backend=backend, v0=v0[b])
        tr = torch.zeros(T, n, device=dev)
        single.forward(T, x=None if x is None else x[:, b].to(dev), traj=tr, out_var=_abi.RP_OUT_SOURCE)
        for a, c, name in ((traj[:, b], tr, "trajectory"), (bp.v[b], single.v, "v"), (bp.source[b], single.source, "source")):
            a, c = a.cpu().numpy(), c.cpu().numpy()
            same = (a.view(np.uint32) == c.view(np.uint32)) | ((a == 0) & (c == 0))
            assert same.all(), f"{model} n={n} B={B} trajectory {b} {name}: {np.count_nonzero(~same)} differ"
    assert np.abs(np.diff(traj.cpu().numpy(), axis=0)).max() > 0


@pytest.mark.gpu
def test_batched_hip_equals_oracle(oracle_backend, hip_device):
    """The kernel against the oracle directly (ragged row count, partial last chunk, state-variable output)."""
    n, B, T = 1036, 4, 30
    W, params = network(n, "li_tanh", seed=3)
    g = torch.Generator().manual_seed(9)
    v0 = torch.randn(B, n, generator=g)
    x = torch.randn(T, B, n, generator=g) * 0.5
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        bp = BatchedPopulation(Population("li_tanh", W, params, 1e-2, dev, backend=backend), B, mfma=False)
        bp.v.copy_(v0.to(dev))
        bp.refresh()
        traj = torch.zeros(T, B, n, device=dev)
        bp.forward(T, x=x.to(dev), traj=traj)
        res.append((traj.cpu().numpy(), bp.v.cpu().numpy(), bp.source.cpu().numpy()))
    for a, c in zip(*res):
        assert np.array_equal(a.view(np.uint32), c.view(np.uint32))


def test_batched_argument_errors(oracle_backend):
    W, params = network(64, "li_tanh", seed=1)
    pop = Population("li_tanh", W, params, 1e-2, "cpu", backend=oracle_backend)
    with pytest.raises(ValueError):
        BatchedPopulation(pop, 3, mfma=False)
    with pytest.raises(ValueError):
        BatchedPopulation(pop, 33)
    with pytest.raises(ValueError, match="exact=True"):   # (5..7 trajectories have no bit-identical form: no silent matrix-core run)
        BatchedPopulation(pop, 6, exact=True)
    assert not BatchedPopulation(pop, 8, exact=True).mfma and BatchedPopulation(pop, 6).mfma
    bp = BatchedPopulation(pop, 2)
    with pytest.raises(RuntimeError):
        bp.forward(5, x=torch.zeros(5, 3, 64))
    qif = Population("qif", W.abs(), dict(tau=1.0, k=1.0, eta=-5.0, tau_s=1.0), 1e-3, "cpu", backend=oracle_backend)
    with pytest.raises(NotImplementedError):
        BatchedPopulation(qif, 2)
    odd = Population("li_tanh", torch.zeros(10, 10), dict(tau=1.0, k=1.0, eta=0.0), 1e-2, "cpu", backend=oracle_backend)
    with pytest.raises(ValueError):
        BatchedPopulation(odd, 2)


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_network_run_batch_equals_separate_runs(target):
    """Network.run_batch: B = 3 trajectories (padded to 4 inside) through inp -> Linear -> rnn -> Linear -> out with
    cutoff and buffer-mean sampling equal three separate Network.run calls from the same initial state, bit for bit."""
    from rectipy_amd import Network
    n, m, k, T, B = 64, 3, 2, 40, 3
    rng = np.random.default_rng(5)
    W, W_in, W_out = rng.standard_normal((n, n)) / n ** 0.5, rng.standard_normal((n, m)), rng.standard_normal((k, n))
    x = rng.standard_normal((B, T, m)).astype(np.float32)
    y0 = rng.standard_normal(n).astype(np.float32)

    for out_fn, s, cutoff in (("identity", 3, 4), ("identity", 1, 0), ("identity", 7, 5), ("tanh", 3, 4)):
        def build():
            net = Network(1e-2, **target)
            net.add_func_node("inp", m, "identity")
            net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", weights=W,
                                source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                                output_var="tanh_op/r", node_vars={"all/li_op/v": y0}, dtype=torch.float32)
            net.add_func_node("out", k, out_fn)
            net.add_edge("inp", "rnn", weights=W_in)
            net.add_edge("rnn", "out", weights=W_out)
            return net

        net = build()
        batch = net.run_batch(x, sampling_steps=s, cutoff=cutoff)
        assert len(batch) == B and np.array_equal(net.state["rnn"].cpu().numpy(), y0)  # the network's own state is untouched
        for b in range(B):
            single = build()
            obs = single.run(x[b], sampling_steps=s, cutoff=cutoff, verbose=False, enable_grad=False)
            assert list(obs["steps"]) == list(batch[b]["steps"])
            # buffer means included: the same running sum in step order / count as the step kernel's
            assert np.array_equal(obs.to_numpy("out"), batch[b].to_numpy("out")), (out_fn, s, cutoff, b)
            assert torch.equal(single.state["rnn"], batch[b]["state"])
    # mfma=True: the same three trajectories on the matrix cores (3 of 16 slots) -- equal to rounding, not bit for bit
    net = build()
    fast = net.run_batch(x, sampling_steps=3, cutoff=4, mfma=True)
    for b in range(B):
        assert np.abs(fast[b].to_numpy("out") - batch[b].to_numpy("out")).max() < 2e-5
        assert np.abs((fast[b]["state"] - batch[b]["state"]).cpu().numpy()).max() < 2e-5


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("with_edge", [True, False])
def test_run_batch_samples_in_kernel(target, with_edge, monkeypatch):
    """`run_batch` never stages [B, T, N]: the buffer means are formed inside the batched step kernels (all three forms)
    and the largest buffer it allocates is O(B x samples x N).  sampling_steps 1 / 3 / 100, cutoff > 0, with an input edge
    (folded into the kernel, RP_X_PROJ) and without; 7 trajectories: `exact=True` (vector ALUs, padded to 8) equals `run`
    bit for bit, the default (matrix cores) to fp32 rounding; 35 trajectories: two groups (32 + 3)."""
    from rectipy_amd import Network
    n, m, k, T = 64, 3, 2, 230
    rng = np.random.default_rng(5)
    W, W_in, W_out = rng.standard_normal((n, n)) / n ** 0.5, rng.standard_normal((n, m)), rng.standard_normal((k, n))
    y0 = rng.standard_normal(n).astype(np.float32)

    def build():
        net = Network(1e-2, **target)
        if with_edge:
            net.add_func_node("inp", m, "identity")
        net.add_diffeq_node("rnn", "neuron_model_templates.rate_neurons.leaky_integrator.tanh", weights=W,
                            source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                            output_var="li_op/v", node_vars={"all/li_op/v": y0}, dtype=torch.float32)
        net.add_func_node("out", k, "identity")
        if with_edge:
            net.add_edge("inp", "rnn", weights=W_in)
        net.add_edge("rnn", "out", weights=W_out)
        return net

    biggest = [0]
    real_empty = torch.empty

    def counting_empty(*a, **kw):
        t = real_empty(*a, **kw)
        biggest[0] = max(biggest[0], t.numel())
        return t

    for B in (7, 35):
        x = rng.standard_normal((B, T, m if with_edge else 1)).astype(np.float32)
        for s, cutoff in ((1, 0), (3, 4), (100, 30)):
            S = len([t for t in range(T) if t >= cutoff and t % s == 0])
            singles = [build().run(x[b], sampling_steps=s, cutoff=cutoff, verbose=False, enable_grad=False) for b in range(B)]
            for exact in (True, False):
                biggest[0] = 0
                monkeypatch.setattr(torch, "empty", counting_empty)
                batch = build().run_batch(x, sampling_steps=s, cutoff=cutoff, exact=exact)
                monkeypatch.setattr(torch, "empty", real_empty)
                assert biggest[0] <= max(32 * max(S, 1) * n, T * 32 * max(m, 1)), (B, s, exact, biggest[0])
                for b in range(B):
                    assert list(singles[b]["steps"]) == list(batch[b]["steps"])
                    a, c = singles[b].to_numpy("out"), batch[b].to_numpy("out")
                    if exact and B <= 8:
                        assert np.array_equal(a, c), (B, s, cutoff, b)
                    else:
                        assert a.shape == c.shape and np.abs(a - c).max() < 5e-5, (B, s, cutoff, b, np.abs(a - c).max())


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("form,B", [("valu", 4), ("mfma16", 11), ("mfma16", 27), ("mfma32", 20)])
def test_batched_sampling_matches_unbatched_sampling(target, form, B):
    """BatchedPopulation.forward with in-kernel sampling (buffer means, post-update recordings, cutoff, two calls, the input
    edge as RP_X_PROJ): per trajectory what Population.forward samples for one -- bit for bit on the vector ALUs, to fp32
    rounding on the matrix cores."""
    dev, backend = target["device"], target["backend"]
    n, m, T, s, cutoff = 260, 5, 47, 4, 6
    W, params = network(n, "li_tanh", seed=B)
    g = torch.Generator().manual_seed(3)
    v0 = torch.randn(B, n, generator=g)
    x = torch.randn(T, B, m, generator=g)
    w_in = torch.randn(n, m, generator=g).to(dev)
    S = len([t for t in range(T) if t >= cutoff and t % s == 0])
    bp = BatchedPopulation(Population("li_tanh", W, params, 1e-2, dev, backend=backend), B, form=form)
    bp.v.copy_(v0.to(dev))
    bp.refresh()
    samples, rv, rr = (torch.zeros(B, S, n, device=dev) for _ in range(3))
    idx = bp.forward(20, x=x[:20].to(dev), w_in=w_in, cutoff=cutoff, sampling_steps=s, samples=samples, rec={0: rv, 1: rr})
    idx = bp.forward(T - 20, x=x[20:].to(dev), w_in=w_in, step0=20, cutoff=cutoff, sampling_steps=s, samples=samples,
                     rec={0: rv, 1: rr}, sample_idx=idx)
    assert idx == S
    for b in sorted({0, B // 2, B - 1}):
        single = Population("li_tanh", W, params, 1e-2, dev, backend=backend, v0=v0[b])
        one, r0, r1 = (torch.zeros(S, n, device=dev) for _ in range(3))
        assert single.forward(T, x=x[:, b].to(dev), w_in=w_in, cutoff=cutoff, sampling_steps=s, samples=one, rec={0: r0, 1: r1}) == S
        for a, c, name in ((samples[b], one, "samples"), (rv[b], r0, "rec v"), (rr[b], r1, "rec r")):
            if form == "valu":
                assert torch.equal(a, c), (b, name)
            else:
                assert float((a - c).abs().max()) < 2e-5, (b, name)


@pytest.mark.gpu
@pytest.mark.parametrize("form,B,n", [("valu", 8, 516), ("mfma16", 13, 1036), ("mfma16", 32, 2052), ("mfma32", 32, 1036)])
def test_batched_sampling_hip_equals_oracle(oracle_backend, hip_device, form, B, n):
    """In-kernel sampling of the batched forms against the oracle's mirrors: bit for bit (samples, recordings, state)."""
    m, T, s, cutoff = 3, 21, 3, 2
    W, params = network(n, "li_tanh", seed=n)
    g = torch.Generator().manual_seed(5)
    v0 = torch.randn(B, n, generator=g)
    x = torch.randn(T, B, m, generator=g)
    w_in = torch.randn(n, m, generator=g)
    S = len([t for t in range(T) if t >= cutoff and t % s == 0])
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        bp = BatchedPopulation(Population("li_tanh", W, params, 1e-2, dev, backend=backend), B, form=form)
        bp.v.copy_(v0.to(dev))
        bp.refresh()
        samples, rv, rr = (torch.zeros(B, S, n, device=dev) for _ in range(3))
        idx = bp.forward(9, x=x[:9].to(dev), w_in=w_in.to(dev), cutoff=cutoff, sampling_steps=s, samples=samples, rec={0: rv, 1: rr})
        bp.forward(T - 9, x=x[9:].to(dev), w_in=w_in.to(dev), step0=9, cutoff=cutoff, sampling_steps=s, samples=samples,
                   rec={0: rv, 1: rr}, sample_idx=idx)
        res.append([t.cpu().numpy() for t in (samples, rv, rr, bp.v)])
    for a, c, name in zip(res[0], res[1], ("samples", "rec v", "rec r", "v")):
        assert np.array_equal(a.view(np.uint32), c.view(np.uint32)), name


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("form", ["mfma16", "mfma32"])
@pytest.mark.parametrize("model,n,B,xw", [("li_tanh", 64, 32, "full"), ("li_tanh", 260, 20, "bcast"), ("li_sigmoid", 516, 9, "full"),
                                         ("li_tanh", 1036, 32, "none"), ("li_tanh", 2048, 16, "full"),
                                         ("li_tanh", 40, 3, "full"), ("li_tanh", 1300, 1, "bcast")])
def test_matrix_core_form_matches_unbatched_runs(target, form, model, n, B, xw):
    """Matrix-core forms (up to 32 trajectories per pass over W, their own summation orders): every trajectory equals an
    unbatched run up to fp32 rounding -- 2e-5 absolute on O(1) trajectories of 25 steps."""
    dev, backend = target["device"], target["backend"]
    W, params = network(n, model, seed=n + B)
    g = torch.Generator().manual_seed(17)
    T = 25
    v0 = torch.randn(B, n, generator=g)
    x = {"full": torch.randn(T, B, n, generator=g) * 0.5, "bcast": torch.randn(T, B, 1, generator=g), "none": None}[xw]
    pop = Population(model, W, params, 1e-2, dev, backend=backend)
    bp = BatchedPopulation(pop, B, form=form)
    assert bp.mfma and bp.form == form
    if 8 < B <= 16:
        assert BatchedPopulation(pop, B).form == BatchedPopulation.DEFAULT_MFMA == "mfma16"
    if B > 16:  # small populations keep the r02 form for more than 16 trajectories (32-row workgroups fill the chip)
        assert BatchedPopulation(pop, B).form == ("mfma32" if n < 12288 else "mfma16")
    bp.v.copy_(v0.to(dev))
    bp.refresh()
    traj = torch.zeros(T, B, n, device=dev)
    bp.forward(10, x=None if x is None else x[:10].to(dev), traj=traj[:10], out_var=_abi.RP_OUT_SOURCE)
    bp.forward(T - 10, x=None if x is None else x[10:].to(dev), traj=traj[10:], out_var=_abi.RP_OUT_SOURCE)
    for b in sorted({0, min(1, B - 1), B // 2, B - 1}):
        single = Population(model, W, params, 1e-2, dev, backend=backend, v0=v0[b])
        tr = torch.zeros(T, n, device=dev)
        single.forward(T, x=None if x is None else x[:, b].to(dev), traj=tr, out_var=_abi.RP_OUT_SOURCE)
        assert float((traj[:, b] - tr).abs().max()) <= 2e-5 and float((bp.v[b] - single.v).abs().max()) <= 2e-5
        assert float((bp.source[b] - single.source).abs().max()) <= 2e-5
    assert np.abs(np.diff(traj.cpu().numpy(), axis=0)).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["mfma16", "mfma32"])
@pytest.mark.parametrize("n,B,sigmoid", [(1036, 32, False), (640, 13, True), (2052, 32, False), (64, 16, False), (60, 5, False),
                                         (1280, 17, False), (4100, 8, True)])
def test_matrix_core_form_hip_equals_oracle(oracle_backend, hip_device, form, n, B, sigmoid):
    """The MFMA kernels against the oracle's mirrors of their fmaf chains: bit for bit (ragged last tile, rows that are
    not a multiple of the workgroup's row block, unused slots, a single tile, more tiles than ring stages)."""
    model = "li_sigmoid" if sigmoid else "li_tanh"
    W, params = network(n, model, seed=n)
    g = torch.Generator().manual_seed(9)
    T = 12
    v0 = torch.randn(B, n, generator=g)
    x = torch.randn(T, B, n, generator=g) * 0.5
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        bp = BatchedPopulation(Population(model, W, params, 1e-2, dev, backend=backend), B, form=form)
        bp.v.copy_(v0.to(dev))
        bp.refresh()
        traj = torch.zeros(T, B, n, device=dev)
        bp.forward(T, x=x.to(dev), traj=traj)
        res.append((traj.cpu().numpy(), bp.v.cpu().numpy().copy(), bp.source.cpu().numpy().copy()))
    for a, c, name in zip(res[0], res[1], ("traj", "v", "source")):
        same = (a.view(np.uint32) == c.view(np.uint32)) | ((a == 0) & (c == 0))
        assert same.all(), f"{name}: {np.count_nonzero(~same)} of {a.size} differ, max {np.abs(a - c).max()}"


@pytest.mark.gpu
def test_mfma16_tile_width_and_ring_depth_keep_the_bits(hip_device, measure_backend):
    """The 16x16x4 form's summation order is defined on 16-column groups, not on the kernel's tile width or LDS ring
    depth: every (tile columns, stages) instance of the measurement build gives the bits of the product library's
    kernel (n with a ragged tail, both slot-block counts)."""
    lib = measure_backend.lib
    n, T = 2308, 6
    W, params = network(n, "li_tanh", seed=3)
    g = torch.Generator().manual_seed(4)
    try:
        for B in (11, 27):
            v0 = torch.randn(B, n, generator=g)
            x = torch.randn(T, B, 1, generator=g)
            outs = []
            # (256-column tiles with two slot blocks need 192 KiB of LDS: that instance does not exist on this chip)
            for tc, stages, waves in ((None, 0, 0), (0, 0, 0), (64, 2, 4), (64, 3, 4), (64, 4, 4), (128, 2, 4), (128, 3, 4)) + \
                    (((256, 2, 4),) if B <= 16 else ((64, 2, 8), (64, 3, 8), (128, 2, 8), (64, 2, 16), (64, 3, 16), (64, 4, 16))):
                if tc is None:  # the product library
                    backend = None
                else:
                    lib.rp_set_batched_mfma_tuning(tc, stages, waves)
                    backend = measure_backend
                bp = BatchedPopulation(Population("li_tanh", W, params, 1e-2, hip_device, backend=backend), B, form="mfma16")
                bp.v.copy_(v0.to(hip_device))
                bp.refresh()
                traj = torch.zeros(T, B, n, device=hip_device)
                bp.forward(T, x=x.to(hip_device), traj=traj)
                outs.append((traj.clone(), bp.v.clone()))
            for tr, v in outs[1:]:
                assert torch.equal(tr, outs[0][0]) and torch.equal(v, outs[0][1])
    finally:
        lib.rp_set_batched_mfma_tuning(0, 0, 0)
