"""The per-step path (user loops calling ``net.forward``; ``fit_bptt`` with recorded variables; truncated BPTT) chains
1-step autograd windows through the state handle; it must give the same gradients as the fused window."""
import numpy as np
import pytest
import torch

from rectipy_amd import Network

TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
TARGETS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture
def target(request, oracle_backend):
    if request.param == "oracle":
        return dict(device="cpu", backend=oracle_backend)
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return dict(device="cuda:0", backend=None)


def build(target, rng_seed=0):
    rng = np.random.default_rng(rng_seed)
    n, k = 24, 2
    W, W_out = rng.standard_normal((n, n)) / np.sqrt(n), rng.standard_normal((k, n)) / np.sqrt(n)
    net = Network(1e-2, **target)
    node = net.add_diffeq_node("rnn", TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in",
                               input_var="li_op/I_ext", output_var="tanh_op/r", dtype=torch.float32,
                               train_params=["weights", "li_op/tau", "li_op/eta"],
                               node_vars={"all/li_op/v": rng.standard_normal(n)})
    net.add_func_node("out", k, "identity")
    edge = net.add_edge("rnn", "out", weights=W_out, train="gd")
    x = rng.standard_normal((60, n)).astype(np.float32) * 0.5
    tgt = rng.standard_normal((60, k)).astype(np.float32) * 0.1
    return net, node, edge, x, tgt


def params_of(node, edge):
    return [p.detach().cpu().numpy().copy() for p in (node["weights"], node["li_op/tau"], node["li_op/eta"], edge.weights)]


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_stepwise_fit_equals_fused_fit(target):
    """fit_bptt three ways: fused windows; fused windows WITH recorded variables of the population (r05: they come out of the window
    the autograd node stores -- before: the per-step walk); the per-step walk (forced).  Same parameters, same records."""
    import warnings
    net_a, node_a, edge_a, x, tgt = build(target)
    net_a.fit_bptt(x, tgt, optimizer="sgd", lr=1e-2, update_steps=20, sampling_steps=5, verbose=False)
    rec = [("rnn", "li_op/v", False), ("rnn", "tanh_op/r", False)]
    net_b, node_b, edge_b, _, _ = build(target)
    with warnings.catch_warnings():
        warnings.simplefilter("error")     # (no fall-back to the walk, which would warn)
        obs_b = net_b.fit_bptt(x, tgt, optimizer="sgd", lr=1e-2, update_steps=20, sampling_steps=5, verbose=False, record_vars=rec)
    net_c, node_c, edge_c, _, _ = build(target)
    net_c._force_stepwise = True
    obs_c = net_c.fit_bptt(x, tgt, optimizer="sgd", lr=1e-2, update_steps=20, sampling_steps=5, verbose=False, record_vars=rec)
    for a, b, c in zip(params_of(node_a, edge_a), params_of(node_b, edge_b), params_of(node_c, edge_c)):
        assert np.abs(a - b).max() == 0 and np.abs(a - c).max() < 2e-6
    for var in (("rnn", "li_op/v"), ("rnn", "tanh_op/r")):
        got, ref = obs_b.to_numpy(var), obs_c.to_numpy(var)
        assert got.shape == ref.shape == (12, 24) and np.abs(got - ref).max() < 2e-6 and np.abs(ref).max() > 0.1
    assert np.abs(obs_b.to_numpy("out") - obs_c.to_numpy("out")).max() < 2e-6
    assert np.abs(obs_b.to_numpy("loss") - obs_c.to_numpy("loss")).max() < 2e-6 and len(obs_b["loss"]) == 12


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_user_loop_with_autograd_and_truncation(target):
    net, node, edge, x, tgt = build(target, rng_seed=1)
    dev = target["device"]
    xs, ts = torch.tensor(x, device=dev), torch.tensor(tgt, device=dev)
    net.compile()
    preds = torch.stack([net.forward(xs[t]) for t in range(30)])        # drop-in per-step loop under autograd
    loss = torch.nn.functional.mse_loss(preds, ts[:30])
    loss.backward()
    g_loop = node["weights"].grad.clone()
    net2, node2, edge2, _, _ = build(target, rng_seed=1)
    traj = node2._grad_window(30, xs[:30])
    loss2 = torch.nn.functional.mse_loss(traj @ edge2.weights.T, ts[:30])
    loss2.backward()
    assert abs(loss.item() - loss2.item()) < 1e-7
    assert (g_loop - node2["weights"].grad).abs().max() < 1e-6 * max(1.0, float(g_loop.abs().max()))
    # truncated BPTT through run(): gradients stop at the truncation boundary
    net3, node3, edge3, _, _ = build(target, rng_seed=1)
    obs = net3.run(xs[:30], sampling_steps=1, verbose=False, enable_grad=True, truncate_steps=10)
    out = obs.stacked("out")
    torch.nn.functional.mse_loss(out[20:], ts[20:30]).backward()       # only the last segment contributes
    net4, node4, edge4, _, _ = build(target, rng_seed=1)
    with torch.no_grad():
        net4.run(xs[:20], sampling_steps=1, verbose=False, enable_grad=False)
    seg = node4._grad_window(10, xs[20:30])
    torch.nn.functional.mse_loss(seg @ edge4.weights.T, ts[20:30]).backward()
    assert (node3["weights"].grad - node4["weights"].grad).abs().max() < 1e-6


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_recorded_variables_of_a_spiking_population_in_fused_fit_windows(target):
    """The same for a QIF population (spikes inside the windows): records of v and s from the stored window equal the walk's."""
    import warnings
    QIF = "neuron_model_templates.spiking_neurons.qif.qif"
    rng = np.random.default_rng(2)
    n, k, T = 20, 2, 60
    W = np.abs(rng.standard_normal((n, n))) * 2.0 / n
    W_out = rng.standard_normal((k, n)) / np.sqrt(n)
    eta = rng.uniform(2.0, 6.0, n)
    x = (rng.standard_normal((T, n)) * 0.5 + 2.0).astype(np.float32)
    tgt = (rng.standard_normal((T, k)) * 0.1).astype(np.float32)
    rec = [("rnn", "qif_op/v", False), ("rnn", "qif_op/s", False)]
    res = []
    for force in (False, True):
        net = Network(2e-2, **target)
        net.add_diffeq_node("rnn", QIF, input_var="qif_op/I_ext", output_var="qif_op/s", weights=W, source_var="qif_op/s",
                            target_var="qif_op/s_in", spike_var="qif_op/spike", reset_var="qif_op/v", spike_threshold=5.0, spike_reset=-5.0,
                            node_vars={"all/qif_op/eta": eta}, dtype=torch.float32, train_params=["weights", "qif_op/eta"])
        net.add_func_node("out", k, "identity")
        edge = net.add_edge("rnn", "out", weights=W_out, train="gd")
        net._force_stepwise = force
        with warnings.catch_warnings():
            if not force:
                warnings.simplefilter("error")
            obs = net.fit_bptt(x, tgt, optimizer="sgd", lr=1e-3, update_steps=20, sampling_steps=3, verbose=False, record_vars=rec)
        res.append((obs.to_numpy(("rnn", "qif_op/v")), obs.to_numpy(("rnn", "qif_op/s")), obs.to_numpy("out"),
                    net.get_node("rnn")["weights"].detach().cpu().numpy().copy(), edge.weights.detach().cpu().numpy().copy()))
    assert res[0][0].shape == (20, n) and np.abs(res[0][1]).max() > 0      # (spikes happened: s moved)
    for a, b in zip(res[0], res[1]):
        assert a.shape == b.shape and np.abs(a - b).max() < 5e-5 * max(1.0, np.abs(b).max())
