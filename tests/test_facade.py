"""Host-side API parity with the reference's own tests (rectipy_tests/test_nodes.py, test_network.py),
run with the CPU oracle doing the arithmetic so that no GPU is needed.  What those tests assert about
construction, naming, errors, reset/detach and simulation bookkeeping is asserted here."""
import numpy as np
import pytest
import torch

from rectipy_amd import Network, Observer, RateNet, SpikeResetNet
from rectipy_amd.edges import Linear

TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
QIF = "neuron_model_templates.spiking_neurons.qif.qif"
IN, OUT, SRC, TGT = "li_op/I_ext", "tanh_op/r", "tanh_op/r", "li_op/r_in"


@pytest.fixture
def where(oracle_backend):
    return dict(device="cpu", backend=oracle_backend)


def test_node_construction_variants(where):
    """mirrors test_2_1_ratenet_init / test_3_1_diffeq_nodes"""
    n = 10
    W = np.random.default_rng(0).standard_normal((n, n))
    rnn2 = RateNet.from_pyrates(TANH, weights=W, source_var=SRC, target_var=TGT, input_var=IN, output_var=OUT,
                                clear=True, verbose=False, dtype=torch.float32, **where)
    rnn3 = SpikeResetNet.from_pyrates(QIF, weights=W, source_var="qif_op/s", target_var="qif_op/s_in",
                                      input_var="qif_op/I_ext", output_var="qif_op/s", reset_var="qif_op/v",
                                      spike_var="qif_op/spike", spike_threshold=1e3, spike_reset=-1e3, clear=True,
                                      verbose=False, dtype=torch.float32, **where)
    rnn4 = RateNet.from_pyrates(TANH, weights=W, source_var=SRC, target_var=TGT, input_var=IN, output_var=OUT,
                                train_params=["weights"], dtype=torch.float32, **where)
    assert len(rnn2.y) == n and len(rnn3.y) == 2 * n
    assert len(list(rnn4.parameters())) - len(list(rnn2.parameters())) == 1
    assert rnn4["li_op/v"].shape[0] == n
    assert rnn3.y.dtype == torch.float32
    assert rnn4.y.requires_grad and not rnn2.y.requires_grad
    with pytest.raises(KeyError):
        _ = rnn2["li_op/u"]
    with pytest.raises(FileNotFoundError):
        RateNet.from_pyrates("neuron_model_templates.rate_neurons.freaky_integrator.tanh", weights=W, source_var=SRC,
                             target_var=TGT, input_var=IN, output_var=OUT, **where)
    with pytest.raises(AttributeError):
        RateNet.from_pyrates("neuron_model_templates.rate_neurons.leaky_integrator.tan", weights=W, source_var=SRC,
                             target_var=TGT, input_var=IN, output_var=OUT, **where)
    with pytest.raises(KeyError):
        RateNet.from_pyrates(TANH, weights=W, source_var="x", target_var=TGT, input_var=IN, output_var=OUT, **where)
    with pytest.raises(NotImplementedError):
        RateNet.from_pyrates("neuron_model_templates.spiking_neurons.qif.qif_reset", weights=W, source_var="s",
                             target_var="s_in", input_var="I_ext", output_var="s", **where)

    nets = [Network(1e-2, **where) for _ in range(4)]
    nets[0].add_diffeq_node("n1", node=TANH, input_var=IN, output_var=OUT, weights=W, source_var=SRC, target_var=TGT)
    nets[1].add_diffeq_node("n1", TANH, weights=W, input_var="I_ext", output_var=OUT, source_var=SRC,
                            target_var="r_in", op="li_op")
    nets[2].add_diffeq_node("n1", TANH, weights=W, input_var=IN, output_var=OUT, source_var=SRC, target_var=TGT,
                            train_params=["weights"])
    nets[3].add_diffeq_node("n1", QIF, weights=W, input_var="I_ext", output_var="s", source_var="s",
                            target_var="s_in", op="qif_op", spike_var="spike", reset_var="v", dtype=torch.float32)
    assert isinstance(nets[0].get_node("n1"), RateNet) and isinstance(nets[3].get_node("n1"), SpikeResetNet)
    assert isinstance(nets[0]["n1"]["node"], RateNet)
    assert len(nets[1]._var_map) - len(nets[0]._var_map) > 0
    assert len(nets[0].get_node("n1").train_params) == 0 and len(nets[2].get_node("n1").train_params) == 1
    assert nets[1].get_var("n1", var="v").shape[0] == n
    net5 = Network(1e-2, **where)
    net5.add_node("n1", rnn2, node_type="diff_eq")
    assert net5.get_node("n1") is rnn2


def test_forward_reset_and_shape_errors(where):
    """mirrors test_2_3_forward / test_2_4_reset"""
    n = 10
    rng = np.random.default_rng(1)
    W, y0 = rng.standard_normal((n, n)), rng.standard_normal(n)
    rnn = RateNet.from_pyrates(TANH, weights=W, source_var=SRC, target_var=TGT, input_var=IN, output_var="li_op/v",
                               node_vars={"all/li_op/v": y0, "all/li_op/tau": 1.0}, dtype=torch.float32, **where)
    x = torch.randn(n)
    r1 = rnn.forward(x)
    assert r1.shape[0] == n and np.allclose(r1.numpy(), y0, atol=1e-6)  # pre-update output
    r2 = rnn.forward(x)
    rnn.reset(y0)
    r3 = rnn.forward(x)
    rnn.reset(y0[0:3], idx=np.arange(0, 3))
    r5 = rnn.forward(x)
    rnn.reset(y0 + 1.0)
    r4 = rnn.forward(x)
    for a, b in ((r1, r2), (r1, r4), (r1, r5)):
        assert np.mean(np.abs(a.numpy() - b.numpy())) > 0
    assert torch.equal(r1, r3)
    assert torch.equal(r1[0:3], r5[0:3]) and np.allclose(r2[3:].numpy(), r5[3:].numpy(), atol=1e-6)
    with pytest.raises(RuntimeError):
        rnn.forward(torch.randn(n + 1))
    with pytest.raises(RuntimeError):
        rnn.forward(np.random.randn(n))
    with pytest.raises(RuntimeError):
        rnn.reset(np.random.randn(n + 1))
    with pytest.raises(RuntimeError):
        rnn.reset(y0, idx=np.arange(0, n + 1))
    # scalar / length-1 inputs broadcast (documentation/interfaces/torch_integration.py:118-120)
    rnn.reset(y0)
    a = rnn.forward(0.3)
    rnn.reset(y0)
    b = rnn.forward(torch.tensor([0.3]))
    assert torch.equal(a, b)
    # set_param / get item
    rnn.set_param("li_op/tau", 5.0)
    assert float(rnn["li_op/tau"][0]) == 5.0
    with pytest.raises(KeyError):
        rnn.set_param("li_op/nope", 1.0)


def test_detach_flags(where):
    """mirrors test_2_2_detach"""
    W = np.random.default_rng(2).standard_normal((10, 10))
    rnn1 = RateNet.from_pyrates(TANH, weights=W, source_var=SRC, target_var=TGT, input_var=IN, output_var=OUT,
                                train_params=["weights"], dtype=torch.float32, **where)
    rnn2 = RateNet.from_pyrates(TANH, weights=W, source_var=SRC, target_var=TGT, input_var=IN, output_var=OUT,
                                train_params=["weights"], dtype=torch.float32, **where)
    rnn2.detach(detach_params=True, requires_grad=False)
    assert rnn1.y.requires_grad and not rnn2.y.requires_grad
    assert rnn1["weights"].requires_grad and not rnn2["weights"].requires_grad


def test_edges_and_parameters(where):
    """mirrors test_edges.py Linear checks and test_3_3 / test_3_5 parameter counts"""
    lin = Linear(4, 3, weights=np.ones((4, 3)))
    assert tuple(lin.weights.shape) == (3, 4) and not lin.weights.requires_grad
    lin2 = Linear(4, 3, detach=False)
    assert lin2.weights.requires_grad and len(list(lin2.parameters())) == 1
    ref = torch.nn.Linear(4, 3, bias=False)
    lin3 = Linear(4, 3, weights=ref.weight.detach(), dtype=torch.float32)
    xin = torch.randn(4)
    assert torch.allclose(lin3(xin), ref(xin), atol=1e-6)
    with pytest.raises(ValueError):
        Linear(4, 3, weights=np.ones((5, 3)))
    with pytest.raises(RuntimeError):
        lin3(torch.randn(5))

    n = 10
    W = np.random.default_rng(3).standard_normal((n, n))
    net = Network(1e-2, **where)
    net.add_diffeq_node("rnn", TANH, weights=W, input_var=IN, output_var=OUT, source_var=SRC, target_var=TGT,
                        train_params=["weights"], dtype=torch.float32)
    net.add_func_node("inp", 3, "tanh")
    net.add_func_node("out", 2, "softmax")
    net.add_edge("inp", "rnn", train="gd")
    net.add_edge("rnn", "out", weights=np.random.randn(2, n))
    assert len(list(net.parameters())) == 2
    assert net.get_edge("rnn", "out").weights.shape == (2, n)
    with pytest.raises(ValueError):
        net.add_edge("rnn", "out", train="bogus")
    with pytest.raises(ValueError):
        net.add_func_node("bad", 3, "relu6")
    net.compile()
    assert net.n_in == 3 and net.n_out == 2
    out = net.forward(torch.randn(3))
    assert out.shape == (2,) and abs(float(out.sum()) - 1.0) < 1e-5
    net.pop_edge("rnn", "out")
    net.pop_node("out")
    assert len(net) == 2


def test_compile_errors(where):
    """mirrors test_3_4: zero or two output nodes raise ValueError"""
    n = 6
    W = np.zeros((n, n))
    net = Network(1e-2, **where)
    net.add_diffeq_node("rnn", TANH, weights=W, input_var=IN, output_var=OUT, source_var=SRC, target_var=TGT)
    net.add_func_node("o1", 2, "identity")
    net.add_func_node("o2", 2, "identity")
    net.add_edge("rnn", "o1")
    net.add_edge("rnn", "o2")
    with pytest.raises(ValueError):
        net.compile()
    net.pop_node("o2")
    net.compile()
    net.add_edge("o1", "rnn", weights=np.zeros((n, 2)))  # a cycle: no node without input edges
    with pytest.raises(ValueError):
        net.compile()


def test_run_matches_manual_forward_loop(where):
    """mirrors test_3_6_simulation: run(sampling_steps=2) == manual forward loop with buffer-mean; recorded
    variables == get_var right after each forward"""
    n, steps, dt = 10, 100, 1e-2
    rng = np.random.default_rng(4)
    W = rng.standard_normal((n, n))
    x = torch.tensor(rng.standard_normal((steps, n)), dtype=torch.float32)
    nets = []
    for _ in range(3):
        net = Network(dt=dt, **where)
        net.add_diffeq_node("rnn", TANH, weights=W, input_var=IN, output_var=OUT, source_var=SRC, target_var=TGT,
                            dtype=torch.float32, record_vars=["li_op/v"])
        nets.append(net)
    res1 = nets[0].run(inputs=x, sampling_steps=2, verbose=False)
    res2 = nets[1].run(inputs=x, record_output=False, record_vars=[("rnn", "li_op/v", False)], verbose=False)
    nets[2].compile()
    res3, res4, buffer = [], [], []
    for step in range(steps):
        out = nets[2].forward(x[step, :])
        buffer.append(out.detach().numpy())
        if step % 2 == 0:
            res3.append(np.mean(buffer, axis=0))
            buffer = []
        res4.append(nets[2].get_var("rnn", var="li_op/v").detach().numpy().copy())
    assert np.abs(res1.to_dataframe("out").values - np.asarray(res3)).max() < 1e-6
    assert np.abs(res2.to_dataframe(("rnn", "li_op/v")).values - np.asarray(res4)).max() < 1e-6
    assert "out" not in res2.recorded_variables


def test_readout_training_recovers_weights(where):
    """mirrors test_3_7_optimization (bptt and ridge legs): a trained readout recovers W_out"""
    dt, n, n_out, steps = 1e-2, 10, 3, 100
    rng = np.random.default_rng(5)
    W, W_out, x = rng.standard_normal((n, n)), rng.standard_normal((n_out, n)), rng.standard_normal((steps, n))
    net = Network(dt, **where)
    net.add_diffeq_node("rnn", TANH, input_var=IN, output_var=OUT, weights=W, source_var=SRC, target_var=TGT,
                        dtype=torch.float32)
    net.add_func_node("output", n_out, "identity")
    net.add_edge("rnn", "output", weights=W_out, train=None)
    y0 = {k: v.clone() for k, v in net.state.items()}
    target = net.run(x, sampling_steps=1, enable_grad=False, verbose=False).to_numpy("out")
    net.pop_edge("rnn", "output")
    net.add_edge("rnn", "output", weights=rng.standard_normal((n_out, n)), train="gd")
    net.reset(y0)
    epochs = 300
    obs = net.fit_bptt([x] * epochs, [target] * epochs, sampling_steps=1, verbose=False, lr=0.1, optimizer="adam")
    fitted = net.get_edge("rnn", "output").weights.detach().numpy()
    assert np.mean((W_out - fitted) ** 2) < 0.5
    assert len(obs["epoch_loss"]) == epochs and obs["epoch_loss"][-1] < obs["epoch_loss"][0]
    net.reset(y0)
    net.pop_node("output")
    obs_r = net.fit_ridge(x, target, sampling_steps=1, add_readout_node=False, verbose=False, alpha=1e-6)
    assert np.mean((W_out - obs_r["w_out"].numpy().T) ** 2) < 0.5
    # Network.test: frozen run + loss against targets (network.py:898-944)
    net.reset(y0)
    rates = net.run(x, sampling_steps=1, enable_grad=False, verbose=False).to_numpy("out")
    net.reset(y0)
    obs_t, loss = net.test(x, rates, sampling_steps=1, verbose=False)
    assert loss == 0.0 and len(obs_t["out"]) == steps
    net.reset(y0)
    _, loss_off = net.test(x, rates + 1.0, sampling_steps=1, verbose=False)
    assert loss_off == pytest.approx(1.0, rel=1e-5)


def test_observer_api():
    obs = Observer(dt=0.1, record_vars=[("a", "v", False), ("a", "w", True)])
    for s in range(3):
        obs.record(s, torch.ones(2) * s, 0.5, [torch.arange(4.0), torch.arange(4.0)])
    obs.record_block([3, 4], torch.ones(2, 2) * 7, [0.1, 0.2], [torch.zeros(2, 4), torch.ones(2, 4)])
    assert obs["steps"] == [0, 1, 2, 3, 4] and len(obs["out"]) == 5
    assert obs.to_numpy("out").shape == (5, 2) and obs.to_numpy(("a", "v")).shape == (5, 4)
    assert obs.to_numpy(("a", "w")).shape == (5,) and obs.to_numpy("loss")[-1] == pytest.approx(0.2)
    assert obs.stacked("out").shape == (5, 2)
    df = obs.to_dataframe("out")
    assert df.shape == (5, 2) and df.index[-1] == pytest.approx(0.4)
    obs.save("extra", [torch.zeros(1)])
    assert obs["extra"][0].shape == (1,) and obs.recorded_state_variables == [("a", "v"), ("a", "w")]
    with pytest.raises(KeyError):
        _ = obs["missing"]


def test_float64_request_is_held_as_float32(where):
    """Documented difference: the reference defaults to float64 node state; the engine computes in float32."""
    import warnings
    n = 6
    W = np.random.default_rng(9).standard_normal((n, n))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = Network(1e-2, dtype=torch.float64, **where)
        net.add_diffeq_node("rnn", TANH, weights=W, input_var=IN, output_var=OUT, source_var=SRC, target_var=TGT)
        net.add_func_node("out", 2, "identity")
        edge = net.add_edge("rnn", "out", weights=np.ones((2, n)))
    assert net.dtype is torch.float32 and edge.weights.dtype is torch.float32
    assert net.get_node("rnn").y.dtype is torch.float32
    obs = net.run(np.zeros((5, n)), verbose=False, enable_grad=False)
    assert obs.to_numpy("out").shape == (5, 2) and obs.to_numpy("out").dtype == np.float32


def test_ridge_solve_primal_and_dual_agree():
    """fit_ridge solves the S x S dual system when there are fewer samples than neurons; same read-out weights."""
    from rectipy_amd import Network
    g = torch.Generator().manual_seed(0)
    for S, N in ((40, 12), (12, 40)):
        X = torch.randn(S, N, generator=g, dtype=torch.float64)
        Y = torch.randn(S, 3, generator=g, dtype=torch.float64)
        w = Network._ridge_solve(X, Y, 1e-2)
        ref = torch.inverse(X.T @ X + 1e-2 * torch.eye(N, dtype=torch.float64)) @ X.T @ Y   # reference network.py:767
        assert w.shape == (N, 3) and (w - ref).abs().max() < 1e-9 * max(1.0, float(ref.abs().max()))


def test_state_writes_reach_the_device_state_of_a_trained_node(where):
    """With train_params the node's `y` is the autograd handle of the window; Network.set_var on a state variable and
    reset() must still change the integration state (the reference's `y` IS the state, nodes.py:166-170)."""
    n = 12
    r = np.random.default_rng(3)
    W, y1 = r.standard_normal((n, n)) / n ** 0.5, r.standard_normal(n).astype(np.float32)
    x = torch.tensor(r.standard_normal((4, n)), dtype=torch.float32)
    outs = []
    for how in ("reset", "set_var"):
        net = Network(1e-2, **where)
        net.add_diffeq_node("rnn", TANH, input_var=IN, output_var="li_op/v", weights=W, source_var=SRC, target_var=TGT,
                            train_params=["weights"], dtype=torch.float32)
        net.compile()
        net.forward(x[0])
        assert net.state["rnn"].requires_grad
        if how == "reset":
            net.reset({"rnn": y1})
        else:
            net.set_var("rnn", "li_op/v", y1)
        outs.append(torch.stack([net.forward(x[t]) for t in range(1, 4)]).detach())
        assert np.array_equal(outs[-1][0].numpy(), y1)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("enable_grad", [True, False])
@pytest.mark.parametrize("train", [None, ["weights"], ["weights", "li_op/tau"]])
def test_in_place_write_into_net_state_is_part_of_the_state(where, train, enable_grad):
    """`net.run(x[:5]); net.state["rnn"][:] = v; net.run(x[5:])`: the reference hands out the live `y`
    (network.py:88-98) and its next step starts from whatever it holds (nodes.py:166-170) -- with or without trainable
    parameters (where `y` is the autograd handle of the last window), with or without a graph being recorded."""
    n = 40
    r = np.random.default_rng(7)
    W = r.standard_normal((n, n)) / n ** 0.5
    v = torch.tensor(r.standard_normal(n), dtype=torch.float32)
    x = torch.tensor(r.standard_normal((9, n)) * 0.3, dtype=torch.float32)
    net = Network(1e-2, **where)
    net.add_diffeq_node("rnn", TANH, input_var=IN, output_var="li_op/v", weights=W, source_var=SRC, target_var=TGT,
                        train_params=train, dtype=torch.float32)
    net.run(x[:5], verbose=False, enable_grad=enable_grad)
    net.state["rnn"][:] = v      # plain write, no no_grad: legal on the reference's y after a step, and here
    obs = net.run(x[5:], verbose=False, enable_grad=enable_grad)
    assert np.array_equal(obs.to_numpy("out")[0], v.numpy()), "the write was dropped"
    # the rest of the run continues from v: same numbers as a fresh network reset to v
    net2 = Network(1e-2, **where)
    net2.add_diffeq_node("rnn", TANH, input_var=IN, output_var="li_op/v", weights=W, source_var=SRC, target_var=TGT,
                         dtype=torch.float32)
    net2.reset({"rnn": v})
    ref = net2.run(x[5:], verbose=False, enable_grad=False)
    assert np.array_equal(obs.to_numpy("out"), ref.to_numpy("out"))
    # a partial write through the variable view of the state keeps the other entries
    before = net.state["rnn"].detach().clone()
    with torch.no_grad():
        net.get_node("rnn")["li_op/v"][:3] = 0.25
    out = net.run(x[:1], verbose=False, enable_grad=enable_grad).to_numpy("out")[0]
    assert np.array_equal(out[:3], np.full(3, 0.25, dtype=np.float32)) and np.array_equal(out[3:], before[3:].numpy())


def test_a_written_state_handle_survives_a_second_read_detach_and_a_subset_reset(where):
    """Found by scripts/fuzz_vs_reference.py: (i) reading `net.state[...]` AGAIN after writing into it must not make the
    node forget the write (the version it compares with is the one of the first hand-out); (ii) `detach()` and (iii) a
    subset `reset(idx=...)` keep the written values -- the reference detaches / clones the very tensor that was written."""
    n = 20
    r = np.random.default_rng(21)
    W = r.standard_normal((n, n)) / n ** 0.5
    x = torch.tensor(r.standard_normal((6, n)) * 0.3, dtype=torch.float32)
    v = torch.tensor(r.standard_normal(n), dtype=torch.float32)
    for how in ("read_again", "detach", "reset_subset"):
        net = Network(1e-2, **where)
        node = net.add_diffeq_node("rnn", TANH, input_var=IN, output_var="li_op/v", weights=W, source_var=SRC,
                                   target_var=TGT, train_params=["weights"], dtype=torch.float32)
        net.run(x[:3], verbose=False)
        net.state["rnn"][:] = v
        want = v.clone()
        if how == "read_again":
            assert torch.equal(net.state["rnn"].detach(), v) and torch.equal(net.state["rnn"].detach(), v)
        elif how == "detach":
            net.detach()
        else:
            node.reset(y=np.array([7.0, -7.0]), idx=np.array([0, 3]))
            want[0], want[3] = 7.0, -7.0
        out = net.run(x[3:], verbose=False).to_numpy("out")
        assert np.array_equal(out[0], want.numpy()), how


def test_gradient_flows_through_a_written_state_handle(where):
    """Autograd sees the in-place write the way it does in the reference: entries overwritten lose their history,
    the others keep it (CopySlices on the handle that the next window starts from)."""
    n = 12
    r = np.random.default_rng(11)
    W = r.standard_normal((n, n)) / n ** 0.5
    x = torch.tensor(r.standard_normal((6, n)) * 0.3, dtype=torch.float32)
    net = Network(1e-2, **where)
    node = net.add_diffeq_node("rnn", TANH, input_var=IN, output_var="li_op/v", weights=W, source_var=SRC,
                               target_var=TGT, train_params=["weights"], dtype=torch.float32)
    net.compile()
    for t in range(3):
        net.forward(x[t])
    net.state["rnn"][: n // 2] = 0.5   # half of the state is replaced by constants
    out = torch.stack([net.forward(x[t]) for t in range(3, 6)])
    assert np.array_equal(out[0, : n // 2].detach().numpy(), np.full(n // 2, 0.5, dtype=np.float32))
    out.sum().backward()
    g = node["weights"].grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0


def test_state_write_on_a_spiking_node_lands_in_the_lagging_copy(where):
    """SpikeResetNet.y is the PRE-update copy the reference rebuilds from _y_start/_y_spike/_y_stop at every step
    (nodes.py:382-392): a write into `net.state[...]` does not reach the integration state there, and does not here
    (reset() is the way to set it).  Pinned against the reference's own class in tests/test_reference_seam.py."""
    n = 16
    r = np.random.default_rng(5)
    W = np.abs(r.standard_normal((n, n))) / n
    x = torch.full((8, n), 15.0)
    outs = []
    for write in (False, True):
        net = Network(1e-3, **where)
        net.add_diffeq_node("qif", QIF, input_var="qif_op/I_ext", output_var="qif_op/s", weights=W,
                            source_var="qif_op/s", target_var="qif_op/s_in", spike_var="qif_op/spike",
                            reset_var="qif_op/v", dtype=torch.float32, node_vars={"all/qif_op/v": -2.0})
        net.run(x[:4], verbose=False, enable_grad=False)
        if write:
            with torch.no_grad():
                net.state["qif"][:] = 1.0
        outs.append(net.run(x[4:], verbose=False, enable_grad=False, record_vars=[("qif", "qif_op/v", False)]))
    assert np.array_equal(outs[0].to_numpy(("qif", "qif_op/v")), outs[1].to_numpy(("qif", "qif_op/v")))


def test_recorded_state_rows_do_not_alias(where):
    """A per-step recorder that keeps the tensors it gets from `node[var]` / `net.state` (what the reference's Observer
    does, observer.py:100-101) sees one distinct state per step."""
    n = 8
    W = np.random.default_rng(5).standard_normal((n, n))
    node = RateNet.from_pyrates(TANH, weights=W, source_var=SRC, target_var=TGT, input_var=IN, output_var="li_op/v",
                                dtype=torch.float32, dt=1e-1, **where)
    x = torch.ones(n)
    kept_v, kept_r, outs = [], [], []
    for _ in range(5):
        outs.append(node.forward(x))
        kept_v.append(node["li_op/v"])
        kept_r.append(node["tanh_op/r"])
    for t in range(4):  # state after step t == output (pre-update state) of step t+1
        assert torch.equal(kept_v[t], outs[t + 1])
        assert torch.allclose(kept_r[t], torch.tanh(outs[t + 1]), atol=1e-6)


def _rate(t, y, I_ext, weights, tau):  # the hand-written field of rectipy_tests/test_nodes.py:32-33
    return -y / tau + I_ext + weights @ torch.tanh(y)


def test_reference_style_constructor_recognises_the_field(where):
    """Seam 2 (reference nodes.py:58-90): RateNet(rnn_func, rnn_args, var_map, param_map) as rectipy_tests/test_nodes.py
    test_2_1 / test_2_3 / test_2_4 use it -- the callable is recognised as the compiled tanh field, a foreign one refused."""
    n = 10
    rng = np.random.default_rng(11)
    W = rng.standard_normal((n, n))
    args = (torch.zeros(n), torch.zeros(n), torch.tensor(W), 1.0)
    rnn1 = RateNet(_rate, args, {"out": [0, n]}, {"in": 0}, dt=1e-3, dtype=torch.float32, **where)
    rnn2 = RateNet.from_pyrates(TANH, weights=W, source_var=SRC, target_var=TGT, input_var=IN, output_var="li_op/v",
                                node_vars={"all/li_op/tau": 1.0}, dt=1e-3, dtype=torch.float32, **where)
    rnn5 = RateNet(_rate, args, {"out": [0, 3]}, {"in": 0}, dtype=torch.float32, **where)
    assert isinstance(rnn1, RateNet) and rnn1.n_out == n and rnn1.n_in == n and rnn5.n_out == 3
    inp = torch.tensor(rng.standard_normal(n), dtype=torch.float32)
    for _ in range(10):
        o1, o2 = rnn1.forward(inp), rnn2.forward(inp)
        assert torch.equal(o1, o2)
    assert rnn5.forward(inp).shape[0] == 3
    # the reference's float64 loop on the same callable
    y = torch.zeros(n, dtype=torch.float64)
    for _ in range(10):
        y = y + 1e-3 * _rate(0, y, inp.double(), torch.tensor(W), 1.0)
    assert float((rnn1.y.double() - y).abs().max()) < 1e-5
    # named parameters: roles are found by perturbation, whatever the names are
    def field(t, y, I_ext, weights, a, b, c):
        return -y / a + c * (weights @ torch.tanh(y)) + I_ext + b
    tau, eta, k = torch.full((n,), 4.0), torch.tensor(rng.standard_normal(n)), torch.tensor(rng.uniform(0.5, 1.5, n))
    node = RateNet(field, (torch.zeros(n), torch.zeros(n), torch.tensor(W), tau, eta, k), {"out": (0, n), "v": (0, n)},
                   {"in": 0, "weights": 1, "leak": 2, "drive": 3, "gain": 4}, train_params=["weights", "gain"],
                   dtype=torch.float32, **where)
    assert node._param_map == {"weights": "weights", "leak": "tau", "drive": "eta", "gain": "k"}
    assert torch.allclose(node["drive"].cpu(), eta.float()) and node["gain"].requires_grad and node.y.requires_grad
    assert len(list(node.parameters())) == 2 and node["v"].shape[0] == n
    with pytest.raises(KeyError):
        _ = node["li_op/u"]
    # reset round trip (test_2_4)
    r1 = rnn1.forward(inp)
    rnn1.reset(np.zeros(n))
    assert torch.equal(rnn1.forward(inp), torch.zeros(n))
    with pytest.raises(RuntimeError):
        rnn1.reset(np.zeros(n + 1))
    # not a member of the compiled family -> refused, no interpreter fallback
    with pytest.raises(NotImplementedError):
        RateNet(lambda t, y, I, w, tau: -y ** 3 / tau + I + w @ torch.tanh(y), args, {"out": [0, n]}, {"in": 0}, **where)
    with pytest.raises(NotImplementedError):
        RateNet(_rate, args, {"out": [0, n]}, {"in": 2}, **where)  # the input slot is not an additive input
    named = RateNet("li_tanh", (torch.zeros(n), torch.zeros(n), torch.tensor(W), 2.0), {"out": [0, n]},
                    {"in": 0, "weights": 1, "li_op/tau": 2}, dtype=torch.float32, **where)
    assert float(named["li_op/tau"][0]) == 2.0


def test_readout_only_fit_skips_the_adjoint_and_gives_the_same_updates(where):
    """Only the read-out edge is trained: no gradient has to pass into the population, so fit_bptt integrates the
    window without the autograd node (no adjoint sweep) -- the read-out updates are those of plain SGD on the collected
    states (two windows; the second starts from the first one's end state)."""
    dt, n, k, T, win, lr = 1e-2, 24, 3, 80, 40, 0.05
    rng = np.random.default_rng(11)
    W, W_out = rng.standard_normal((n, n)) * 0.3, rng.standard_normal((k, n)) * 0.2
    x, target = rng.standard_normal((T, n)).astype(np.float32) * 0.5, rng.standard_normal((T, k)).astype(np.float32)

    def build(train):
        net = Network(dt, **where)
        net.add_diffeq_node("rnn", TANH, input_var=IN, output_var="li_op/v", weights=W, source_var=SRC, target_var=TGT,
                            dtype=torch.float32)
        net.add_func_node("out", k, "identity")
        net.add_edge("rnn", "out", weights=W_out.copy(), train=train)
        return net

    # states of the population over the run: re-run a bare population network whose output IS the state
    bare = Network(dt, **where)
    bare.add_diffeq_node("rnn", TANH, input_var=IN, output_var="li_op/v", weights=W, source_var=SRC, target_var=TGT,
                         dtype=torch.float32)
    V = torch.tensor(bare.run(x, sampling_steps=1, enable_grad=False, verbose=False).to_numpy("out"))
    w = torch.tensor(W_out, dtype=torch.float32, requires_grad=True)
    tgt = torch.tensor(target)
    expected_losses = []
    for a in range(0, T, win):
        loss = torch.nn.functional.mse_loss(V[a:a + win] @ w.T, tgt[a:a + win])
        (g,) = torch.autograd.grad(loss, w)
        with torch.no_grad():
            w -= lr * g
        expected_losses.append(float(loss.detach()))

    net = build("gd")
    rnn = net.get_node("rnn")
    calls = []
    inner = rnn._grad_window
    rnn._grad_window = lambda *a, **kw: (calls.append(1), inner(*a, **kw))[1]
    obs = net.fit_bptt(x, target, optimizer="sgd", lr=lr, update_steps=win, sampling_steps=1, verbose=False)
    assert not calls, "a read-out-only fit went through the population's autograd window"
    fitted = net.get_edge("rnn", "out").weights.detach()
    assert torch.allclose(fitted, w.detach(), atol=2e-6)
    assert [float(obs["loss"][b - 1]) for b in (win, T)] == pytest.approx(expected_losses, rel=1e-5)


@pytest.mark.parametrize("readout", [False, True])
def test_sampled_outputs_have_the_same_bits_with_and_without_gradient_tracking(where, readout):
    """A sample is the mean of the outputs buffered since the previous one (network.py:590-597).  The fused no-grad run
    forms it in the step kernel (sequential float32 sum, true division); the gradient-tracking run, the read-out path and
    the step-by-step walk form it with torch ops -- in the same order and with a true division (Network._seq_mean), not
    with torch.mean, whose order and reciprocal multiplication are the device's: found by the HIP-vs-oracle fuzz as a
    last-bit difference between the CPU and the GPU at sampling_steps = 3."""
    rng = np.random.default_rng(3)
    n, T = 37, 29
    W = rng.standard_normal((n, n)) / 4
    x = rng.standard_normal((T, n)).astype(np.float32)
    outs = []
    for grad, stepwise in ((False, False), (True, False), (False, True)):
        net = Network(1e-2, **where)
        net._force_stepwise = stepwise
        net.add_diffeq_node("rnn", TANH, weights=W, source_var=SRC, target_var=TGT, input_var=IN, output_var="li_op/v",
                            train_params=["weights"] if grad else None, dtype=torch.float32)
        if readout:
            net.add_func_node("out", 3, "tanh")
            net.add_edge("rnn", "out", weights=np.random.default_rng(4).standard_normal((3, n)))
        outs.append(net.run(x, sampling_steps=3, cutoff=2, verbose=False, enable_grad=grad).to_numpy("out"))
    assert outs[0].shape[0] == len([t for t in range(T) if t >= 2 and t % 3 == 0])
    assert np.array_equal(outs[0], outs[1])
    # (the walk applies the read-out per step as a matrix-vector product, the windows as one matrix-matrix product: BLAS rounding)
    assert np.array_equal(outs[0], outs[2]) if not readout else np.abs(outs[0] - outs[2]).max() < 1e-6
    # and they are the sequential sums: three consecutive per-step outputs, added in order, divided by three
    net = Network(1e-2, **where)
    net.add_diffeq_node("rnn", TANH, weights=W, source_var=SRC, target_var=TGT, input_var=IN, output_var="li_op/v", dtype=torch.float32)
    if readout:
        net.add_func_node("out", 3, "tanh")
        net.add_edge("rnn", "out", weights=np.random.default_rng(4).standard_normal((3, n)))
    per_step = net.run(x, sampling_steps=1, verbose=False, enable_grad=False).to_numpy("out")
    want = ((per_step[4] + per_step[5]) + per_step[6]) / np.float32(3.0)
    assert np.array_equal(outs[0][1], want) and np.array_equal(outs[0][0], (per_step[2] + per_step[3]) / np.float32(2.0))
