"""Surrogate-gradient BPTT through the spiking population (rp_qif_backward) against
(a) the reference's autograd through SpikeResetNet + Spike.backward (golden G10, float64 truth) and
(b) the CPU oracle, bit for bit, on the GPU."""
import os

import numpy as np
import pytest
import torch

from rectipy_amd import Network, SpikeResetNet, _abi
from rectipy_amd.engine import Population

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


def qif_node(g, target, train):
    n = g["W"].shape[0]
    return SpikeResetNet.from_pyrates(
        QIF, input_var="qif_op/I_ext", output_var="qif_op/s", weights=g["W"], source_var="qif_op/s",
        target_var="qif_op/s_in", spike_var="qif_op/spike", reset_var="qif_op/v",
        node_vars={"all/qif_op/eta": g["eta"], "all/qif_op/tau": g["tau"], "all/qif_op/k": g["k"],
                   "all/qif_op/tau_s": g["tau_s"], "all/qif_op/v": g["y0"][:n], "all/qif_op/s": g["y0"][n:]},
        spike_threshold=float(g["theta"]), spike_reset=float(g["v_reset"]), dt=float(g["dt"]), dtype=torch.float32,
        train_params=train, **target)


TRAIN = ["weights", "qif_op/tau", "qif_op/k", "qif_op/eta", "qif_op/tau_s"]


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g10_gradients_match_reference_autograd(target):
    g = dict(np.load(os.path.join(GOLD, "g10_qif_bptt.npz")))
    dev = target["device"]
    node = qif_node(g, target, TRAIN)
    assert abs(node._spike_slope - float(g["slope"])) < 1e-12
    node.enable_spike_recording(4096)
    W_out = torch.tensor(g["W_out"], dtype=torch.float32, device=dev, requires_grad=True)
    x = torch.tensor(g["x"], dtype=torch.float32, device=dev)
    tgt = torch.tensor(g["target"], dtype=torch.float32, device=dev)
    traj = node._grad_window(x.shape[0], x)
    assert len(node.spikes()) >= 10, "the window must contain spikes for this test to mean anything"
    loss = torch.nn.functional.mse_loss(traj @ W_out.T, tgt)
    assert abs(loss.item() - g["loss_f64"][-1]) < 1e-5
    loss.backward()
    lr = float(g["lr"])
    for key, got, before in (("W", node["weights"].grad, g["W"]), ("tau", node["qif_op/tau"].grad, g["tau"]),
                             ("k", node["qif_op/k"].grad, g["k"]), ("eta", node["qif_op/eta"].grad, g["eta"]),
                             ("tau_s", node["qif_op/tau_s"].grad, g["tau_s"]), ("W_out", W_out.grad, g["W_out"])):
        ref = (before - g[f"{key}_after_f64"]) / lr
        err = np.abs(got.cpu().numpy().astype(np.float64) - ref).max()
        assert np.abs(ref).max() > 0
        assert err <= 1e-3 * np.abs(ref).max(), f"grad {key}: err {err} vs scale {np.abs(ref).max()}"


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g10_fit_bptt_through_network(target):
    """The same window driven by Network.fit_bptt (fused chain: rnn -> Linear -> identity)."""
    g = dict(np.load(os.path.join(GOLD, "g10_qif_bptt.npz")))
    net = Network(float(g["dt"]), **target)
    node = qif_node(g, target, TRAIN)
    net.add_node("rnn", node, node_type="diff_eq")
    net.add_func_node("out", g["W_out"].shape[0], "identity")
    edge = net.add_edge("rnn", "out", weights=g["W_out"].copy(), train="gd")
    lr = float(g["lr"])
    obs = net.fit_bptt(g["x"], g["target"], optimizer="sgd", lr=lr, update_steps=g["x"].shape[0], sampling_steps=1,
                       verbose=False)
    assert np.abs(obs.to_numpy("out") - g["pred_f64"]).max() < 1e-4
    assert abs(obs.to_numpy("loss")[-1] - g["loss_f64"][-1]) < 1e-5
    for key, val, before in (("W", node["weights"], g["W"]), ("eta", node["qif_op/eta"], g["eta"]),
                             ("W_out", edge.weights, g["W_out"])):
        ref_grad = (before - g[f"{key}_after_f64"]) / lr
        got_grad = (before - val.detach().cpu().numpy().astype(np.float64)) / lr
        floor = np.abs(before).max() * 1.2e-7 / lr
        assert np.abs(got_grad - ref_grad).max() <= 2e-3 * np.abs(ref_grad).max() + floor, key


@pytest.mark.gpu
@pytest.mark.parametrize("N,T", [(24, 40), (256, 30), (1000, 12)])
@pytest.mark.parametrize("grad_var,xw", [(_abi.RP_OUT_SOURCE, "full"), (_abi.RP_OUT_STATE0, "bcast")])
@pytest.mark.parametrize("transposed", [False, True])
def test_qif_backward_bit_exact(oracle_backend, hip_device, N, T, grad_var, xw, transposed):
    gen = torch.Generator().manual_seed(N + T)
    W = torch.rand(N, N, generator=gen) * 20.0 / N
    params = dict(tau=torch.rand(N, generator=gen) * 0.4 + 0.8, k=torch.rand(N, generator=gen) * 0.4 + 0.8,
                  eta=torch.randn(N, generator=gen) - 5.0, tau_s=torch.rand(N, generator=gen) + 0.5)
    v0 = torch.rand(N, generator=gen) * 65.0 - 5.0
    s0 = torch.rand(N, generator=gen)
    x = 15.0 + torch.randn(T, N if xw == "full" else 1, generator=gen)
    g_out = torch.randn(T, N, generator=gen) * 0.1
    lv0, ls0 = torch.randn(N, generator=gen) * 0.01, torch.randn(N, generator=gen) * 0.01
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        pop = Population("qif", W, params, 1e-3, dev, v0=v0, s0=s0, theta=100.0, v_reset=-100.0, backend=backend)
        pop.transposed_adjoint = transposed
        tv, ts, ta, ut, gx = (torch.zeros(T, N, device=dev) for _ in range(5))
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        ev = torch.zeros(N * T, 2, dtype=torch.int32, device=dev)
        pop.forward(T, x=x.to(dev), save_v=tv, save_src=ts, save_acc=ta, spike_events=ev, spike_count=cnt)
        lv, ls = lv0.clone().to(dev), ls0.clone().to(dev)
        gt, gk, ge, gts = (torch.zeros(N, device=dev) for _ in range(4))
        keep = pop.backward(tv, ts, g_out.to(dev), grad_var, lv, ut, traj_acc=ta, grad_tau=gt, grad_k=gk, grad_eta=ge,
                            grad_x=gx, lam_s=ls, grad_tau_s=gts, slope=0.5, x=x.to(dev))
        if dev != "cpu":
            torch.cuda.synchronize()
        res.append((lv, ls, ut, gt, gk, ge, gts, gx, int(cnt.item())))
        del keep
    assert res[0][-1] == res[1][-1] and res[0][-1] > 0
    names = ("lambda_v", "lambda_s", "u_traj", "grad_tau", "grad_k", "grad_eta", "grad_tau_s", "grad_x")
    for a, b, name in zip(res[0][:-1], res[1][:-1], names):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        same = (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))
        assert same.all(), f"{name}: {np.count_nonzero(~same)} of {a.size} differ (max abs {np.abs(a - b).max()})"


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("transposed", [True, False], ids=["transposed-copy", "column-strips"])
@pytest.mark.parametrize("n", [512, 1028])
def test_g32_qif_gradients_at_kernel_sizes_match_reference_autograd(target, n, transposed):
    """Golden G32 (tests/golden/make_golden_qif_bptt_large.py): one SGD window of the reference's fit_bptt through its
    SpikeResetNet at N = 512 / 1028, all parameters trained.  Both adjoint forms of the spiking sweep run here with full
    256-column chunks (and the ragged 4-column chunk at 1028).  Inputs are regenerated from tests/golden/recipes.py and
    checked against the stored fingerprints first.  Tolerances: 4e-4 of each gradient's largest entry (measured: <= 9.3e-5,
    which is also what the reference's own float32 run deviates from its float64 run); column sums of dW 1e-3."""
    import sys
    sys.path.insert(0, GOLD)
    import recipes
    from make_golden_qif_bptt_large import network
    g = dict(np.load(os.path.join(GOLD, f"g32_qif_bptt_n{n}.npz")))
    T, kout, dev = int(g["T"]), 3, target["device"]
    P = network(n, 3200 + n)
    W_out0 = (recipes.normal((kout, n), 3250 + n) / np.float32(np.sqrt(n))).astype(np.float32)
    x = (15.0 + recipes.normal((T, n), 3260 + n)).astype(np.float32)
    tgt = (0.1 * recipes.normal((T, kout), 3270 + n)).astype(np.float32)
    for name, arr in (("W", P["W"]), ("x", x), ("target", tgt), ("y0", P["y0"])):
        assert np.array_equal(recipes.fingerprint(arr), g["fp_" + name]), f"regenerated {name} differs from the generator's"
    node = qif_node(dict(theta=g["theta"], v_reset=g["v_reset"], dt=g["dt"], **P), target, TRAIN)
    node._pop.transposed_adjoint = transposed
    node.enable_spike_recording(1 << 16)
    W_out = torch.tensor(W_out0, device=dev, requires_grad=True)
    traj = node._grad_window(T, torch.tensor(x, device=dev))
    assert len(node.spikes()) >= n // 4, "the window must contain many spikes for this test to mean anything"
    pred = traj @ W_out.T
    assert np.abs(pred.detach().cpu().numpy() - g["pred_f64"]).max() < 2e-5
    loss = torch.nn.functional.mse_loss(pred, torch.tensor(tgt, device=dev))
    assert abs(loss.item() - float(g["loss_f64"])) < 2e-6
    loss.backward()
    gW = node["weights"].grad.cpu().numpy().astype(np.float64)
    scale = np.abs(g["gW_rows_f64"]).max()
    assert np.abs(gW[g["rows"]] - g["gW_rows_f64"]).max() <= 4e-4 * scale
    assert np.abs(gW.sum(axis=0) - g["gW_colsum_f64"]).max() <= 1e-3 * np.abs(g["gW_colsum_f64"]).max()
    assert abs(recipes.fingerprint(gW)[1] - g["gW_fp_f64"][1]) <= 1e-4 * g["gW_fp_f64"][1]      # sum of squares of all of it
    for name, key in (("tau", "qif_op/tau"), ("k", "qif_op/k"), ("eta", "qif_op/eta"), ("tau_s", "qif_op/tau_s")):
        ref = g[f"g{name}_f64"]
        assert np.abs(node[key].grad.cpu().numpy() - ref).max() <= 4e-4 * np.abs(ref).max(), name
    assert np.abs(W_out.grad.cpu().numpy() - g["gW_out_f64"]).max() <= 1e-3 * np.abs(g["gW_out_f64"]).max()
