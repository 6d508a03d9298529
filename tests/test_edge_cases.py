"""Edge cases of the path: the Spike function grid (G7, incl. x == 0), empty windows, tiny and ragged
populations, sampling corner cases, argument errors."""
import os

import numpy as np
import pytest
import torch

from rectipy_amd import Network, RateNet, _abi
from rectipy_amd.engine import Population

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


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g7_spike_function_grid(target):
    """Spike.forward = heaviside with value 1 at 0; Spike.backward = g / (1 + slope |x|)^2 (nodes.py:468-481):
    one QIF step whose v - theta sits on the reference's grid exposes both through the kernels."""
    g = dict(np.load(os.path.join(GOLD, "g7_spike_fn.npz")))
    grid, slope = g["grid"], float(g["slope"])
    n, dev = len(grid), target["device"]
    theta = 100.0
    v0 = torch.tensor(theta + grid, dtype=torch.float32)
    pop = Population("qif", torch.zeros(n, n), dict(tau=1.0, k=1.0, eta=-5.0, tau_s=1.0), 1e-3, dev, v0=v0, s0=0.0,
                     theta=theta, v_reset=-100.0, backend=target["backend"])
    tv, ts, ut = (torch.zeros(1, n, device=dev) for _ in range(3))
    pop.forward(1, save_v=tv, save_src=ts)
    spikes = pop.source_local.cpu().numpy()          # s' = s (1 - dt/tau_s) + spike with s = 0
    assert np.array_equal(spikes, g["fwd_f32"]) and np.array_equal(spikes, (grid >= 0).astype(np.float32))
    lam_v, lam_s = torch.zeros(n, device=dev), torch.ones(n, device=dev)
    keep = pop.backward(tv, ts, None, _abi.RP_OUT_SOURCE, lam_v, ut, lam_s=lam_s, slope=slope)
    # with lambda_v = 0 the new lambda_v is exactly lambda_s * surrogate(v - theta)
    x32 = (v0 - torch.tensor(theta)).numpy()  # the kernels see v - theta in float32
    expect = 1.0 / (1.0 + slope * np.abs(x32.astype(np.float64))) ** 2
    assert np.abs(lam_v.cpu().numpy() - expect).max() < 1e-6
    assert np.abs(lam_v.cpu().numpy() - g["bwd_f64"]).max() < 2e-5  # grid points are rounded to fp32 around theta
    del keep


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("n", [1, 3, 4, 7, 8, 260])
def test_tiny_and_ragged_populations(target, n):
    rng = np.random.default_rng(n)
    W, y0 = rng.standard_normal((n, n)) / np.sqrt(n), rng.standard_normal(n)
    node = RateNet.from_pyrates(TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in",
                                input_var="li_op/I_ext", output_var="li_op/v", node_vars={"all/li_op/v": y0},
                                dt=1e-2, dtype=torch.float32, **target)
    T = 30
    x = torch.tensor(rng.standard_normal((T, n)), dtype=torch.float32, device=target["device"])
    traj = torch.empty(T, n, device=target["device"])
    node.run_window(T, x=x, traj=traj)
    v = y0.copy()
    outs = []
    for t in range(T):
        outs.append(v.copy())
        v = v + 1e-2 * (-v / 10.0 + W @ np.tanh(v) + x[t].cpu().numpy().astype(np.float64))
    assert np.abs(traj.cpu().numpy() - np.asarray(outs)).max() < 1e-5
    assert np.abs(node.y.cpu().numpy() - v).max() < 1e-5


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_empty_and_corner_sampling(target):
    n = 16
    W = np.random.default_rng(0).standard_normal((n, n)) / 4
    net = Network(1e-2, **target)
    net.add_diffeq_node("rnn", TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in",
                        input_var="li_op/I_ext", output_var="li_op/v", dtype=torch.float32)
    before = net.state["rnn"].clone()
    obs = net.run(np.zeros((0, n), dtype=np.float32), verbose=False)       # no steps at all
    assert obs["steps"] == [] and torch.equal(net.state["rnn"], before)
    obs = net.run(np.ones((5, n), dtype=np.float32), cutoff=10, verbose=False)  # everything cut off
    assert obs["steps"] == [] and len(obs["out"]) == 0 and not torch.equal(net.state["rnn"], before)
    obs = net.run(np.ones((7, n), dtype=np.float32), sampling_steps=100, verbose=False)  # one sample, at step 0
    assert obs["steps"] == [0] and obs.to_numpy("out").shape == (1, n)
    node = net.get_node("rnn")
    node.reset(np.zeros(n))
    obs = net.run(np.ones((9, n), dtype=np.float32), sampling_steps=4, cutoff=3, verbose=False)
    assert obs["steps"] == [4, 8]   # first buffer holds steps 3,4 ; second 5..8
    node.reset(np.zeros(n))
    traj = torch.empty(9, n, device=target["device"])
    node.run_window(9, x=torch.ones(9, n, device=target["device"]), traj=traj)
    assert np.allclose(obs.to_numpy("out")[0], traj[3:5].mean(0).cpu().numpy(), atol=1e-7)
    assert np.allclose(obs.to_numpy("out")[1], traj[5:9].mean(0).cpu().numpy(), atol=1e-7)


def test_argument_errors(oracle_backend):
    where = dict(device="cpu", backend=oracle_backend)
    W = np.zeros((6, 6))
    kw = dict(weights=W, source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext", dtype=torch.float32)
    with pytest.raises(KeyError):
        RateNet.from_pyrates(TANH, output_var="li_op/w", **kw, **where)
    with pytest.raises(KeyError):
        RateNet.from_pyrates(TANH, output_var="li_op/v", train_params=["li_op/nope"], **kw, **where)
    with pytest.raises(RuntimeError):
        RateNet.from_pyrates(TANH, output_var="li_op/v", **{**kw, "weights": np.zeros((6, 5))}, **where)
    with pytest.raises(ValueError):
        RateNet.from_pyrates(TANH, output_var="li_op/v", weights=W, input_var="li_op/I_ext", dtype=torch.float32, **where)
    node = RateNet.from_pyrates(TANH, output_var="li_op/v", **kw, **where)
    with pytest.raises(RuntimeError):
        node.run_window(4, x=torch.zeros(3, 6))      # fewer input rows than steps
    with pytest.raises(RuntimeError):
        node.run_window(4, x=torch.zeros(4, 5))      # wrong width
    net = Network(1e-2, **where)
    net.add_node("rnn", node, node_type="diff_eq")
    with pytest.raises(ValueError):
        net.fit_bptt(np.zeros((4, 6)), np.zeros((5, 6)), verbose=False)
    with pytest.raises(ValueError):
        net.fit_bptt(np.zeros((4, 6)), np.zeros((4, 6)), optimizer="nope", verbose=False)
    with pytest.raises(ValueError):
        net.fit_bptt(np.zeros((4, 6)), np.zeros((4, 6)), loss="nope", verbose=False)
    with pytest.raises(ValueError):  # no edge was added with train="rls"
        net.fit_rls(np.zeros((4, 6)), np.zeros((4, 6)))
    with pytest.raises(NotImplementedError):
        net.fit_eprop(np.zeros((4, 6)), np.zeros((4, 6)))


def test_g7_stand_alone_spike_function():
    """rectipy_amd.Spike (the torch function the reference exposes as rectipy.nodes.Spike) on the reference's grid."""
    from rectipy_amd import Spike
    g = dict(np.load(os.path.join(GOLD, "g7_spike_fn.npz")))
    old = Spike.slope
    Spike.slope = float(g["slope"])
    try:
        x = torch.tensor(g["grid"], dtype=torch.float64, requires_grad=True)
        y = Spike.apply(x)
        y.sum().backward()
        assert np.array_equal(y.detach().numpy(), g["fwd_f64"])
        assert np.abs(x.grad.numpy() - g["bwd_f64"]).max() < 1e-12
    finally:
        Spike.slope = old
