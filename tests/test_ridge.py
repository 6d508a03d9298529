"""Network.fit_ridge + Network.test against the reference's own run (golden G26, tests/golden/make_golden_ridge.py;
reference rectipy/network.py:709-784, 898-944): state collection through the fused window, the closed-form solve, the
read-out edge it adds, and the test pass through that edge.

The reference solves (X^T X + alpha I) w = X^T Y in the network dtype; in float32 its own answer is dominated by
rounding (G26: cond = 1.5e6, the float32 reference deviates from its float64 run by 0.02 in w_out and 0.08 in y), so the
float64 reference run is the truth here.  The engine collects X in float32 (1e-6 from the reference's states) and solves
in float64 (torch): tolerances below are for that combination."""
import os

import numpy as np
import pytest
import torch

from rectipy_amd import Network

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


def err(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.abs(a.astype(np.float64) - np.asarray(b, dtype=np.float64)).max()


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g26_fit_ridge_then_test(target):
    g = dict(np.load(os.path.join(GOLD, "g26_fit_ridge.npz")))
    n, m = g["W"].shape[0], g["W_in"].shape[1]
    net = Network(float(g["dt"]), **target)
    net.add_func_node("inp", m, "identity")
    net.add_diffeq_node("rnn", TANH, output_var="li_op/v", weights=g["W"], source_var="tanh_op/r",
                        target_var="li_op/r_in", input_var="li_op/I_ext", dtype=torch.float32,
                        node_vars={"all/li_op/tau": g["tau"], "all/li_op/k": g["k"], "all/li_op/eta": g["eta"],
                                   "all/li_op/v": g["y0"]})
    net.add_edge("inp", "rnn", weights=g["W_in"])
    obs = net.fit_ridge(g["x"], g["target"], sampling_steps=1, alpha=float(g["alpha"]), verbose=False)
    assert err(obs.to_numpy("out"), g["X_f64"]) <= 2e-5, "collected states"
    w_out, y = obs["w_out"], obs["y"]
    assert tuple(w_out.shape) == g["w_out_f64"].shape and tuple(y.shape) == g["y_f64"].shape
    assert err(w_out, g["w_out_f64"]) <= 2e-3 * np.abs(g["w_out_f64"]).max(), f"w_out {err(w_out, g['w_out_f64'])}"
    assert err(y, g["y_f64"]) <= 2e-4 * np.abs(g["y_f64"]).max(), f"y {err(y, g['y_f64'])}"
    assert err(net.state["rnn"], g["state_fit_f64"]) <= 2e-5
    # the read-out node and edge the reference adds (network.py:778-780)
    assert "readout" in net.nodes and tuple(net.get_edge("rnn", "readout").weights.shape) == (g["w_out_f64"].shape[1], n)
    obs2, loss = net.test(g["x2"], g["target2"], loss="mse", sampling_steps=1, verbose=False)
    assert err(obs2.to_numpy("out"), g["test_out_f64"]) <= 1e-3 * np.abs(g["test_out_f64"]).max()
    assert abs(loss - float(g["test_loss_f64"])) <= 1e-3 * float(g["test_loss_f64"])
    assert err(net.state["rnn"], g["state_test_f64"]) <= 2e-5
    # the engine is closer to the float64 truth than the reference's own float32 run
    assert err(y, g["y_f64"]) < 0.1 * err(g["y_f32"], g["y_f64"])


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("key,stride", [("T60_s1", 1), ("T600_s10", 10), ("T120_s2", 2)])
def test_g39_fit_ridge_with_as_many_samples_as_neurons(target, key, stride):
    """The ill-conditioned regime (r04 review): 60 samples for 64 neurons, alpha = 1e-3, cond(X^T X + alpha I) = 3e6 ... 4e7.
    The engine collects float32 states (2e-6 off the reference's float64 ones) and forms the Gram matrix and the solve in
    float64: the read-out weights stay within 5e-4 of max |w_out| of the reference's float64 run -- the bound stated in
    Network.fit_ridge -- where the reference's own float32 run is off by 7e-2 ... 1."""
    g = dict(np.load(os.path.join(GOLD, "g39_fit_ridge_illcond.npz")))
    net = Network(float(g["dt"]), **target)
    net.add_func_node("inp", g["W_in"].shape[1], "identity")
    net.add_diffeq_node("rnn", TANH, output_var="li_op/v", weights=g["W"], source_var="tanh_op/r",
                        target_var="li_op/r_in", input_var="li_op/I_ext", dtype=torch.float32,
                        node_vars={"all/li_op/tau": g["tau"], "all/li_op/k": g["k"], "all/li_op/eta": g["eta"],
                                   "all/li_op/v": g["y0"]})
    net.add_edge("inp", "rnn", weights=g["W_in"])
    obs = net.fit_ridge(g["x_" + key], g["target_" + key], sampling_steps=stride, alpha=float(g["alpha"]), verbose=False)
    ref, ref32 = g[f"w_out_{key}_f64"], g[f"w_out_{key}_f32"]
    assert err(obs.to_numpy("out"), g[f"X_{key}_f64"]) <= 2e-5, "collected states"
    scale = np.abs(ref).max()
    assert err(obs["w_out"], ref) <= 5e-4 * scale, f"w_out {err(obs['w_out'], ref) / scale}"
    assert err(obs["w_out"], ref) < 0.02 * err(ref32, ref)
