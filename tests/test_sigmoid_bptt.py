"""Gradients of the sigmoid population's own parameters (r_max, s, v0 of leaky_integrator.yaml:22) next to W, tau, k,
eta: against the reference's float64 autograd through `RateNet` (golden G22, output v) and against torch autograd over a
float64 restatement of the same loop for output r (the reference's RateNet can only emit state slices)."""
import os

import numpy as np
import pytest
import torch

from rectipy_amd import Network, RateNet

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIGMOID = "neuron_model_templates.rate_neurons.leaky_integrator.sigmoid"
TARGETS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
NAMES = ("tau", "k", "eta")
ACT = ("r_max", "s", "v0")


@pytest.fixture
def target(request, oracle_backend):
    if request.param == "oracle":
        return dict(device="cpu", backend=oracle_backend)
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return dict(device="cuda:0", backend=None)


def node_vars(g):
    nv = {f"all/li_op/{q}": g[q] for q in NAMES}
    nv.update({f"all/sigmoid_op/{q}": g[q] for q in ACT})
    nv["all/li_op/v"] = g["y0"]
    return nv


TRAIN = ["weights"] + [f"li_op/{q}" for q in NAMES] + [f"sigmoid_op/{q}" for q in ACT]


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g22_every_sigmoid_parameter_against_reference_autograd(target):
    g = dict(np.load(os.path.join(GOLD, "g22_sigmoid_bptt.npz")))
    net = Network(float(g["dt"]), **target)
    node = net.add_diffeq_node("rnn", SIGMOID, input_var="li_op/I_ext", output_var="li_op/v", weights=g["W"],
                               source_var="sigmoid_op/r", target_var="li_op/r_in", dtype=torch.float32,
                               node_vars=node_vars(g), train_params=TRAIN)
    net.add_func_node("out", g["W_out"].shape[0], "identity")
    edge = net.add_edge("rnn", "out", weights=g["W_out"].copy(), train="gd")
    lr, T = float(g["lr"]), g["x"].shape[0]
    obs = net.fit_bptt(g["x"], g["target"], optimizer="sgd", lr=lr, update_steps=T, sampling_steps=1, verbose=False)
    assert abs(obs["loss"][-1] - g["loss_f64"][-1]) < 2e-5 * max(1.0, abs(g["loss_f64"][-1]))
    got = {"W": node["weights"], "W_out": edge.weights}
    got.update({q: node[f"li_op/{q}"] for q in NAMES})
    got.update({q: node[f"sigmoid_op/{q}"] for q in ACT})
    for key, after in got.items():
        before = g[key]
        ref = (before - g[f"{key}_after_f64"]) / lr               # gradient implied by the reference's SGD step
        mine = (before - after.detach().cpu().numpy().astype(np.float64)) / lr
        scale = np.abs(ref).max()
        assert scale > 0, key
        floor = np.abs(before).max() * 1.2e-7 / lr                  # fp32 parameter rounding of my own SGD step
        assert np.abs(mine - ref).max() <= 2e-3 * scale + floor, f"grad {key}: {np.abs(mine - ref).max()} vs {scale}"


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_output_r_gradients_against_torch_autograd(target):
    """output_var = sigmoid_op/r: dL/dr_t also reaches r_max, s, v0 directly through the loss."""
    n, T, dt = 24, 60, 1e-2
    gen = torch.Generator().manual_seed(3)
    rnd = lambda *shape: torch.randn(*shape, generator=gen, dtype=torch.float64)
    W = rnd(n, n) * 1.5 / n ** 0.5
    P = dict(tau=torch.rand(n, generator=gen, dtype=torch.float64) * 5 + 5, k=torch.rand(n, generator=gen, dtype=torch.float64) + 0.5,
             eta=rnd(n) * 0.1, r_max=torch.rand(n, generator=gen, dtype=torch.float64) + 0.5,
             s=torch.rand(n, generator=gen, dtype=torch.float64) * 2 + 0.5, v0=rnd(n) * 0.2)
    y0, x, G = rnd(n), rnd(T, n) * 0.5, rnd(T, n) * 0.1

    # float64 restatement with autograd: L = sum_t <G_t, r_t>, r_t = act(v_t) the pre-update output
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    Wl = W.clone().requires_grad_(True)
    v, loss = y0.clone(), 0.0
    for t in range(T):
        r = leaves["r_max"] / (1 + torch.exp(leaves["s"] * (leaves["v0"] - v)))
        loss = loss + (G[t] * r).sum()
        v = v + dt * (-v / leaves["tau"] + leaves["k"] * (Wl @ r) + x[t] + leaves["eta"])
    loss.backward()

    node = RateNet.from_pyrates(SIGMOID, input_var="li_op/I_ext", output_var="sigmoid_op/r", weights=W.numpy(),
                                source_var="sigmoid_op/r", target_var="li_op/r_in", dt=dt, dtype=torch.float32,
                                node_vars={**{f"all/li_op/{q}": P[q].numpy() for q in NAMES},
                                           **{f"all/sigmoid_op/{q}": P[q].numpy() for q in ACT},
                                           "all/li_op/v": y0.numpy()}, train_params=TRAIN, **target)
    dev = target["device"]
    traj = node._grad_window(T, x.to(torch.float32).to(dev))
    (traj * G.to(torch.float32).to(dev)).sum().backward()
    pairs = [("weights", Wl.grad)] + [(f"li_op/{q}", leaves[q].grad) for q in NAMES] + \
            [(f"sigmoid_op/{q}", leaves[q].grad) for q in ACT]
    for name, ref in pairs:
        got = node[name].grad.detach().cpu().double()
        assert (got - ref).abs().max() <= 2e-4 * ref.abs().max() + 1e-7, name
