"""qif_sfa / lif / ik templates: front end + kernels against golden vectors from the reference's SpikeResetNet
(G11-G13; spike events EXACT, synaptic trajectories within fp32 tolerance), and HIP vs oracle bit for bit."""
import os

import numpy as np
import pytest
import torch

from rectipy_amd import Network, SpikeResetNet, _abi
from rectipy_amd.engine import Population

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BASE = "neuron_model_templates.spiking_neurons."
TARGETS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture
def target(request, oracle_backend):
    if request.param == "oracle":
        return dict(device="cpu", backend=oracle_backend)
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return dict(device="cuda:0", backend=None)


def gold(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def check(node, g, T, x, key="", n_state=3, tol_s=1e-3):
    dev = node._pop.device
    n = g["W"].shape[0]
    node.enable_spike_recording(8192)
    traj = torch.empty(T, n, device=dev)
    node.run_window(T, x=x, traj=traj)
    ev = node.spikes()
    ref_ev = g[f"events_{key}f64"]
    assert ref_ev.shape[0] > 50, "fixture without spikes"
    assert (ref_ev == g[f"events_{key}f32"]).all()
    assert ev.shape == ref_ev.shape and (ev == ref_ev).all(), "spike events differ from the reference"
    err = np.abs(traj[::int(g["stride"])].cpu().numpy() - g[f"s_{key}f64"]).max()
    assert err < tol_s, f"s trajectory error {err}"
    assert len(node.y) == n_state * n and len(node.current_state) == n_state * n
    fin = node.current_state.cpu().numpy().astype(np.float64)
    ref = g[f"final_{key}f64"]
    scale = np.maximum(1.0, np.abs(ref))
    assert (np.abs(fin - ref) / scale).max() < 5e-3


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g11_qif_sfa(target):
    g = gold("g11_qif_sfa")
    n, T = g["W"].shape[0], int(g["T"])
    net = Network(float(g["dt"]), **target)
    node = net.add_diffeq_node(
        "sfa", BASE + "qif.qif_sfa", weights=g["W"], source_var="s", target_var="s_in", input_var="I_ext",
        output_var="s", spike_var="spike", reset_var="v", op="qif_sfa_op", dtype=torch.float32,
        node_vars={"eta": g["eta"], "alpha": g["alpha"], "tau_x": g["tau_x"]},
        spike_threshold=float(g["theta"]), spike_reset=float(g["v_reset"]))
    assert isinstance(node, SpikeResetNet) and net.get_var("sfa", "x").shape[0] == n
    x = torch.full((T, 1), float(g["x_const"]), device=target["device"])
    check(node, g, T, x)
    assert float(node["qif_sfa_op/x"].abs().max()) > 0  # adaptation variable moved


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("mode", ["membrane", "synapse"])
def test_g12_lif_both_input_targets(target, mode):
    g = gold("g12_lif")
    T = g["x"].shape[0]
    node = SpikeResetNet.from_pyrates(
        BASE + "lif.lif", input_var="lif_op/I_ext" if mode == "membrane" else "lif_op/s_ext", output_var="lif_op/s",
        weights=g["W"], source_var="lif_op/s", target_var="lif_op/s_in", spike_var="lif_op/spike",
        reset_var="lif_op/v", dt=float(g["dt"]), dtype=torch.float32,
        node_vars={"all/lif_op/tau": g["tau"], "all/lif_op/k": g["k"], "all/lif_op/eta": g["eta"],
                   "all/lif_op/tau_s": g["tau_s"], "all/lif_op/v": g["y0"][:32], "all/lif_op/s": g["y0"][32:]},
        spike_threshold=float(g["theta"]), spike_reset=float(g["v_reset"]), **target)
    x = torch.tensor(g["x"], dtype=torch.float32, device=target["device"])
    check(node, g, T, x, key=mode + "_", n_state=2)


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g13_ik(target):
    g = gold("g13_ik")
    n, T = g["W"].shape[0], int(g["T"])
    names = ("C", "k", "v_r", "v_theta", "eta", "g", "E_r", "b", "tau_u", "kappa", "tau_s")
    node_vars = {f"all/ik_op/{q}": g[q] for q in names}
    node_vars.update({"all/ik_op/v": g["y0"][:n], "all/ik_op/u": g["y0"][n:2 * n], "all/ik_op/s": g["y0"][2 * n:]})
    node = SpikeResetNet.from_pyrates(
        BASE + "ik.ik", input_var="ik_op/I_ext", output_var="ik_op/s", weights=g["W"], source_var="ik_op/s",
        target_var="ik_op/s_in", spike_var="ik_op/spike", reset_var="ik_op/v", dt=float(g["dt"]), dtype=torch.float32,
        node_vars=node_vars, spike_threshold=float(g["theta"]), spike_reset=float(g["v_reset"]), **target)
    x = torch.full((T, 1), float(g["x_const"]), device=target["device"])
    check(node, g, T, x, tol_s=2e-3)
    # state order v, u, s as in the YAML equations
    assert torch.equal(node["ik_op/u"], node.y[n:2 * n])


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["qif_sfa", "lif", "ik"])
def test_new_spiking_models_hip_vs_oracle_bit_exact(oracle_backend, hip_device, model):
    N, T = 512, 400
    gen = torch.Generator().manual_seed(17)
    W = torch.rand(N, N, generator=gen) * 10.0 / N
    if model == "qif_sfa":
        params = dict(tau=1.0, k=1.0, eta=torch.randn(N, generator=gen) - 5.0, tau_s=1.0, alpha=1.0, tau_x=10.0)
        kw = dict(v0=torch.rand(N, generator=gen) * 100 - 20, s0=0.0, aux0=0.1, theta=100.0, v_reset=-100.0)
        dt, x = 1e-3, torch.full((T, 1), 15.0)
    elif model == "lif":
        params = dict(tau=8.0, k=1.0, eta=0.2, tau_s=0.5)
        kw = dict(v0=torch.rand(N, generator=gen), s0=0.0, theta=1.0, v_reset=0.0)
        dt, x = 1e-2, torch.rand(T, N, generator=gen)
    else:
        params = dict(C=100.0, k=0.7, v_r=-60.0, v_theta=-40.0, eta=torch.randn(N, generator=gen) * 5, g=1.0, E_r=0.0,
                      b=-2.0, tau_u=33.33, kappa=10.0, tau_s=6.0)
        kw = dict(v0=torch.rand(N, generator=gen) * 90 - 60, s0=0.0, aux0=0.0, theta=40.0, v_reset=-55.0)
        dt, x = 1e-1, torch.full((T, 1), 120.0)
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        pop = Population(model, W, params, dt, dev, backend=backend, **kw)
        ev = torch.zeros(N * T, 2, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        traj = torch.zeros(T, N, device=dev)
        sa = torch.zeros(T, N, device=dev) if pop.aux is not None else None
        sv = torch.zeros(T, N, device=dev)
        pop.forward(T, x=x.to(dev), traj=traj, out_var=_abi.RP_OUT_SOURCE, save_v=sv, save_aux=sa, spike_events=ev,
                    spike_count=cnt, x_target=_abi.RP_XT_SYNAPSE if model == "lif" else _abi.RP_XT_MEMBRANE)
        c = int(cnt.item())
        e = ev[:c].cpu().numpy()
        e = e[np.lexsort((e[:, 1], e[:, 0]))]
        res.append((c, e, traj.cpu().numpy(), sv.cpu().numpy(), None if sa is None else sa.cpu().numpy(),
                    pop.v.cpu().numpy(), pop.source.cpu().numpy()))
    assert res[0][0] == res[1][0] and res[0][0] > N // 4
    assert (res[0][1] == res[1][1]).all()
    for a, b in zip(res[0][2:], res[1][2:]):
        if a is not None:
            assert ((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))).all()


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("name,op,states,extra", [
    ("g18_iku", "iku_op", ("v", "u", "s"), ("tau_s",)),
    ("g19_ik_biexp", "ik_biexp_op", ("v", "u", "s", "x"), ("tau_r", "tau_d")),
])
def test_g18_g19_ik_population_mean_variants(target, name, op, states, extra):
    """iku / ik_biexp: the recovery variable is driven by mean(v) and mean(spike) over the population
    (ik.yaml:34-68) -- a per-step reduction kernel in front of the step kernel."""
    g = gold(name)
    n, T = g["W"].shape[0], int(g["T"])
    names = ("C", "k", "v_r", "v_theta", "eta", "g", "E_r", "b", "tau_u", "kappa") + extra
    node_vars = {f"all/{op}/{q}": g[q] for q in names}
    for i, st in enumerate(states):
        node_vars[f"all/{op}/{st}"] = g["y0"][i * n:(i + 1) * n]
    node = SpikeResetNet.from_pyrates(
        BASE + "ik." + op[:-3], input_var=f"{op}/I_ext", output_var=f"{op}/s", weights=g["W"], source_var=f"{op}/s",
        target_var=f"{op}/s_in", spike_var=f"{op}/spike", reset_var=f"{op}/v", dt=float(g["dt"]), dtype=torch.float32,
        node_vars=node_vars, spike_threshold=float(g["theta"]), spike_reset=float(g["v_reset"]), **target)
    x = torch.full((T, 1), float(g["x_const"]), device=target["device"])
    check(node, g, T, x, n_state=len(states), tol_s=2e-3)
    # trainable as well (adjoint with the population-mean terms; gradients are pinned by golden G20 / G21)
    trained = SpikeResetNet.from_pyrates(
        BASE + "ik." + op[:-3], input_var=f"{op}/I_ext", output_var=f"{op}/s", weights=g["W"], source_var=f"{op}/s",
        target_var=f"{op}/s_in", spike_var=f"{op}/spike", reset_var=f"{op}/v", dtype=torch.float32,
        train_params=["weights", f"{op}/b"], **target)
    assert len(list(trained.parameters())) == 2


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["iku", "ik_biexp"])
def test_mean_coupled_models_hip_vs_oracle_bit_exact(oracle_backend, hip_device, model):
    N, T = 1500, 300
    gen = torch.Generator().manual_seed(23)
    W = torch.rand(N, N, generator=gen) * 10.0 / N
    params = dict(C=100.0, k=0.7, v_r=-60.0, v_theta=-40.0, eta=torch.randn(N, generator=gen) * 5, g=1.0, E_r=0.0,
                  b=-2.0, tau_u=33.33, kappa=10.0)
    params.update(dict(tau_s=6.0) if model == "iku" else dict(tau_r=2.0, tau_d=6.0))
    kw = dict(v0=torch.rand(N, generator=gen) * 90 - 60, s0=0.0, aux0=0.0, theta=40.0, v_reset=-55.0)
    x = torch.full((T, 1), 120.0)
    res = []
    for backend, dev in ((oracle_backend, "cpu"), (None, hip_device)):
        pop = Population(model, W, params, 1e-1, dev, backend=backend, **kw)
        ev = torch.zeros(N * T, 2, dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        traj = torch.zeros(T, N, device=dev)
        sa = torch.zeros(T, N, device=dev)
        pop.forward(T, x=x.to(dev), traj=traj, out_var=_abi.RP_OUT_SOURCE, save_aux=sa, spike_events=ev, spike_count=cnt)
        c = int(cnt.item())
        e = ev[:c].cpu().numpy()
        res.append((c, e[np.lexsort((e[:, 1], e[:, 0]))], traj.cpu().numpy(), sa.cpu().numpy(), pop.v.cpu().numpy()))
    assert res[0][0] == res[1][0] and res[0][0] > N // 4 and (res[0][1] == res[1][1]).all()
    for a, b in zip(res[0][2:], res[1][2:]):
        assert ((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))).all()
