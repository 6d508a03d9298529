"""MultiSpikeResetNet (reference rectipy/nodes.py:404-465) against golden G29 captured from the reference's class
(tests/golden/make_golden_multispike.py): two QIF populations in one node -- POST-update output, population-major state,
spike detection on a private copy of the reset variables that starts at zero and survives reset()."""
import os

import numpy as np
import pytest
import torch

from rectipy_amd import MultiSpikeResetNet, Network, SpikeResetNet

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
QIF = "neuron_model_templates.spiking_neurons.qif.qif"
TARGETS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture
def target(request, oracle_backend):
    if request.param == "oracle":
        return dict(device="cpu", backend=oracle_backend)
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return dict(device="cuda:0", backend=None)


def err(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.abs(a.astype(np.float64) - np.asarray(b, dtype=np.float64)).max()


def build(g, target, more_vars=None, input_var="qif_op/I_ext", output_var="e/qif_op/s"):
    ne, ni = int(g["ne"]), int(g["ni"])
    return SpikeResetNet.from_pyrates(
        QIF, input_var=input_var, output_var=output_var, weights=g["W"], source_var="qif_op/s",
        target_var="qif_op/s_in", spike_var=["e/qif_op/spike", "i/qif_op/spike"], reset_var=["e/qif_op/v", "i/qif_op/v"],
        populations={"e": ne, "i": ni},
        node_vars={"all/qif_op/eta": g["eta"], "e/qif_op/tau": 1.0, "i/qif_op/tau": 0.8, "all/qif_op/tau_s": g["tau_s"],
                   "e/qif_op/v": g["v0"][:ne], "i/qif_op/v": g["v0"][ne:], "all/qif_op/s": 0.0, **(more_vars or {})},
        spike_threshold=float(g["theta"]), spike_reset=float(g["v_reset"]), dt=float(g["dt"]), dtype=torch.float32,
        **target)


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g29_multispike_matches_reference(target):
    g = dict(np.load(os.path.join(GOLD, "g29_multispike.npz")))
    ne, ni = int(g["ne"]), int(g["ni"])
    node = build(g, target)
    assert isinstance(node, MultiSpikeResetNet) and node.n_out == ne and len(node.y) == 2 * (ne + ni)
    assert err(node.y, g["y0"]) < 1e-6  # population-major layout [v_e, s_e, v_i, s_i]
    assert node.variable_names[:4] == ["e/qif_op/v", "e/qif_op/s", "i/qif_op/v", "i/qif_op/s"]
    x = torch.tensor([float(g["x_const"])], device=target["device"])
    kept, T = set(g["kept"].tolist()), 3000  # the fixture ran 3000 steps before its reset
    outs, ys = [], []
    for step in range(T):
        o = node.forward(x)
        if step in kept:
            outs.append(o.clone())
            ys.append(node.y.clone())
    outs, ys = torch.stack(outs), torch.stack(ys)
    # neuron 3 starts above threshold: the zero-initialised detector must not spike it in step 0 (v keeps growing)
    assert float(ys[0][3]) > 150.0 and float(ys[1][3]) == float(g["v_reset"])
    v_cols = np.r_[0:ne, 2 * ne:2 * ne + ni]
    s_cols = np.r_[ne:2 * ne, 2 * ne + ni:2 * (ne + ni)]
    assert err(outs, g["out_f64"]) <= 5e-4
    assert err(ys[:, s_cols], g["y_f64"][:, s_cols]) <= 5e-4
    # membrane potentials run up to +-100 and beyond between resets: compare relative to their magnitude
    dv = np.abs(ys[:, v_cols].cpu().numpy().astype(np.float64) - g["y_f64"][:, v_cols])
    assert (dv / np.maximum(1.0, np.abs(g["y_f64"][:, v_cols]))).max() <= 2e-3
    # spike timing is identical: the reset pattern (v == v_reset exactly) of every stored row
    assert np.array_equal(ys[:, v_cols].cpu().numpy() == float(g["v_reset"]), g["y_f64"][:, v_cols] == float(g["v_reset"]))
    # reset(): new state, STALE detector
    assert err(node._detector, g["detector_before_reset_f64"]) <= 2e-2
    node.reset(g["y1"])
    after = torch.stack([node.forward(x) for _ in range(3)])
    assert err(after, g["after_reset_f64"]) <= 5e-4
    y_ref = g["y_after_reset_f64"]
    assert (np.abs(node.y.cpu().numpy() - y_ref) / np.maximum(1.0, np.abs(y_ref))).max() <= 2e-3


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_multispike_window_equals_stepping_and_runs_in_a_network(target):
    g = dict(np.load(os.path.join(GOLD, "g29_multispike.npz")))
    ne, ni = int(g["ne"]), int(g["ni"])
    dev = target["device"]
    a, b = build(g, target), build(g, target)
    T = 400
    x = torch.full((T, 1), float(g["x_const"]), device=dev)
    stepped = torch.stack([a.forward(x[t]) for t in range(T)])
    traj = torch.empty(T, ne, device=dev)
    b.run_window(T, x=x, traj=traj)
    assert torch.equal(traj, stepped) and torch.equal(a.y, b.y)
    # inside a Network: evaluated step by step (post-update output timing), read-out on top
    net = Network(float(g["dt"]), **target)
    net.add_node("rnn", build(g, target), node_type="diff_eq")
    net.add_func_node("out", 2, "identity")
    W_out = np.random.default_rng(0).standard_normal((2, ne))
    net.add_edge("rnn", "out", weights=W_out)
    obs = net.run(np.full((T, 1), float(g["x_const"]), dtype=np.float32), sampling_steps=1, verbose=False, enable_grad=False)
    assert err(obs.to_numpy("out"), stepped.cpu().numpy() @ W_out.T) <= 1e-4


def test_multispike_argument_errors(oracle_backend):
    g = dict(np.load(os.path.join(GOLD, "g29_multispike.npz")))
    where = dict(device="cpu", backend=oracle_backend)
    with pytest.raises(ValueError):
        SpikeResetNet.from_pyrates(QIF, input_var="qif_op/I_ext", output_var="qif_op/s", weights=g["W"],
                                   source_var="qif_op/s", target_var="qif_op/s_in",
                                   spike_var=["e/qif_op/spike", "i/qif_op/spike"], reset_var=["e/qif_op/v", "i/qif_op/v"],
                                   dtype=torch.float32, **where)  # populations missing
    with pytest.raises(KeyError):
        build(g, where, more_vars={"x/qif_op/eta": 1.0})


def test_multispike_input_and_output_addressing(oracle_backend):
    """Input into ONE population (zeros elsewhere) and the output variable of EVERY population."""
    g = dict(np.load(os.path.join(GOLD, "g29_multispike.npz")))
    ne, ni = int(g["ne"]), int(g["ni"])
    where = dict(device="cpu", backend=oracle_backend)
    a = build(g, where, input_var="e/qif_op/I_ext", output_var="qif_op/s")
    b = build(g, where, output_var="qif_op/s")
    assert a.n_in == ne and a.n_out == ne + ni and b.n_in == ne + ni
    xe = torch.linspace(5.0, 20.0, ne)
    xfull = torch.cat([xe, torch.zeros(ni)])
    for _ in range(300):
        oa, ob = a.forward(xe), b.forward(xfull)
        assert torch.equal(oa, ob)
    assert torch.equal(a.y, b.y) and oa.shape[0] == ne + ni
    assert torch.equal(oa[:ne], a["e/qif_op/s"]) and torch.equal(oa[ne:], a["i/qif_op/s"])
    with pytest.raises(RuntimeError):
        a.forward(torch.zeros(ne + 1))
