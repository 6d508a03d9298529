"""a ``rectipy_amd`` node plugged into the REFERENCE's own ``rectipy.Network``
(`Network.add_node`, rectipy/network.py:178-211) and driven by the reference's Python time loop, Observer and
`fit_bptt`.  Container-only: needs /root/reference (never shipped); skipped elsewhere, so nothing of the reference
travels.  The arithmetic runs on the CPU oracle backend (there is no GPU in the build container); the host logic under
test -- what `forward`, `y`, `__getitem__`, `reset`, `detach`, `parameters` hand to a foreign loop -- is the product's.

Behaviour to match: rectipy_tests/test_network.py:323-339 (per-step loop + get_var), observer.py:100-101 (the Observer
stores REFERENCES to recorded tensors), nodes.py:166-170 (y is rebound per step)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/rectipy"),
                                reason="the reference is only present in the build container")

TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, GOLD)
    from make_golden import import_reference
    Network, RateNet, SpikeResetNet, Spike = import_reference()
    return Network


def gold(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def amd_node(g, oracle_backend, train=None, output_var="li_op/v"):
    import rectipy_amd
    return rectipy_amd.RateNet.from_pyrates(
        TANH, input_var="li_op/I_ext", output_var=output_var, weights=g["W"], source_var="tanh_op/r",
        target_var="li_op/r_in", node_vars={"all/li_op/tau": g["tau"], "all/li_op/k": g["k"], "all/li_op/eta": g["eta"],
                                            "all/li_op/v": g["y0"]},
        dt=float(g["dt"]), dtype=torch.float32, device="cpu", backend=oracle_backend, train_params=train)


def close(a, b, atol):
    assert np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max() <= atol


@pytest.mark.parametrize("s", [1, 2, 3])
@pytest.mark.parametrize("reduce", [False, True])
def test_reference_network_run_with_amd_node(ref, oracle_backend, s, reduce):
    """G2 through rectipy.Network.run: outputs, steps, RECORDED state variables (one distinct row per sample) and the
    final state."""
    g = gold("g2_run_sampling")
    net = ref(float(g["dt"]), dtype=torch.float32)
    node = amd_node(g, oracle_backend)
    net.add_node("rnn", node, node_type="diff_eq")
    obs = net.run(torch.tensor(g["x"], dtype=torch.float32), sampling_steps=s, cutoff=int(g["cutoff"]), verbose=False,
                  enable_grad=False, record_vars=[("rnn", "li_op/v", reduce)])
    key = f"s{s}_{'red' if reduce else 'full'}"
    assert list(obs["steps"]) == list(g[f"steps_{key}_f64"])
    close(obs.to_numpy("out"), g[f"out_{key}_f64"], 2e-5)
    rec = obs.to_numpy(("rnn", "li_op/v"))
    close(rec, g[f"v_{key}_f64"], 2e-5)
    if not reduce:
        assert len({r.tobytes() for r in rec}) == len(rec), "recorded rows alias one buffer"
    close(net.state["rnn"].detach().numpy(), g[f"state_{key}_f64"], 2e-5)


def test_reference_user_loop_get_var_and_reset(ref, oracle_backend):
    """The per-step user loop of rectipy_tests/test_network.py:323-339: forward, get_var after the step (post-update
    state), tensors kept from earlier steps stay what they were; reset round trip."""
    g = gold("g1_rate_tanh_n16")
    net = ref(float(g["dt"]), dtype=torch.float32)
    node = amd_node(g, oracle_backend)
    net.add_node("rnn", node, node_type="diff_eq")
    net.compile()
    x = torch.tensor(g["x_full"], dtype=torch.float32)
    outs, kept = [], []
    for t in range(x.shape[0]):
        outs.append(net.forward(x[t]))
        kept.append(net.get_var("rnn", "li_op/v"))       # references, as the reference's Observer keeps them
    outs = torch.stack(outs).numpy()
    close(outs, g["out_full_f64"], 2e-5)
    # get_var after step t is y_{t+1} = the output of step t+1
    close(torch.stack(kept[:-1]).numpy(), g["out_full_f64"][1:], 2e-5)
    close(kept[-1].numpy(), g["yT_full_f64"], 2e-5)
    # derived variable handed out per step keeps its value as well
    r0 = net.get_var("rnn", "tanh_op/r").clone()
    r_ref = net.get_var("rnn", "tanh_op/r")
    net.forward(x[0]); net.forward(x[1]); net.forward(x[2])
    assert torch.equal(r0, r_ref)
    # reset round trip (rectipy_tests/test_nodes.py:154-190)
    held = net.state["rnn"]
    before = held.clone()
    net.reset({"rnn": g["y0"]})
    assert torch.equal(held, before), "reset must rebind, not overwrite a handed-out state tensor"
    again = torch.stack([net.forward(x[t]) for t in range(x.shape[0])]).numpy()
    assert np.array_equal(again, outs)


def test_state_write_through_live_tensor(ref, oracle_backend):
    """`net.state[node][:] = y` and `set_var` on a state variable change the integration state, as with the reference's
    node where `y` IS the state (network.py:96, 166-170)."""
    g = gold("g1_rate_tanh_n16")
    outs = []
    for how in ("reset", "live", "set_var"):
        net = ref(float(g["dt"]), dtype=torch.float32)
        node = amd_node(g, oracle_backend)
        net.add_node("rnn", node, node_type="diff_eq", op="li_op")
        net.compile()
        y1 = torch.tensor(g["y0"] + 1.0, dtype=torch.float32)
        x = torch.tensor(g["x_full"], dtype=torch.float32)
        net.forward(x[0])
        if how == "reset":
            net.reset({"rnn": y1})
        elif how == "live":
            with torch.no_grad():
                net.state["rnn"][:] = y1
        else:
            net.set_var("rnn", "li_op/v", y1)   # (the reference's set_var does not relabel, network.py:170-174)
        outs.append(torch.stack([net.forward(x[t]) for t in range(1, 6)]))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(outs[0][0], torch.tensor(g["y0"] + 1.0, dtype=torch.float32))


@pytest.mark.parametrize("enable_grad", [True, False])
@pytest.mark.parametrize("train", [None, ["weights"]])
def test_in_place_state_write_between_two_runs_matches_the_reference(ref, oracle_backend, train, enable_grad):
    """`net.run(x[:5]); net.state["rnn"][:] = v; net.run(x[5:])` inside the reference's own rectipy.Network
    (network.py:88-98 hands out the live y): once with the reference's RateNet around the hand-written field, once with
    the rectipy_amd node -- with train_params the latter's `y` is the autograd handle of the last window, and the
    write must still reach the device state."""
    from make_golden import rate_node
    sys.path.insert(0, "/root/reference")
    from rectipy.nodes import RateNet as RefRateNet
    g = gold("g1_rate_tanh_n16")
    x = torch.tensor(g["x_full"][:10], dtype=torch.float32)
    v = torch.tensor(g["y0"] * 0.5 + 0.3, dtype=torch.float32)
    outs = []
    for which in ("reference", "amd"):
        net = ref(float(g["dt"]), dtype=torch.float32)
        if which == "reference":
            node = rate_node(RefRateNet, g["W"], g["tau"], g["k"], g["eta"], g["y0"], float(g["dt"]), torch.float32,
                             train_params=train)
        else:
            node = amd_node(g, oracle_backend, train=train)
        net.add_node("rnn", node, node_type="diff_eq", op="li_op")
        net.run(x[:5], verbose=False, enable_grad=enable_grad)
        net.state["rnn"][:] = v
        obs = net.run(x[5:], verbose=False, enable_grad=enable_grad)
        outs.append(obs.to_numpy("out"))
    assert np.array_equal(outs[0][0], v.numpy()) and np.array_equal(outs[1][0], v.numpy())
    close(outs[1], outs[0], 2e-6)


def test_state_write_on_spiking_node_is_ignored_by_the_reference_too(ref, oracle_backend):
    """The reference's SpikeResetNet rebuilds `y` from its private segments at every step (nodes.py:382-392), so a
    write into net.state[...] never reaches the integration; rectipy_amd.SpikeResetNet keeps that behaviour."""
    from make_golden import make_qif
    sys.path.insert(0, "/root/reference")
    from rectipy.nodes import SpikeResetNet as RefSpikeNet
    import rectipy_amd
    n, dt = 8, 1e-3
    r = np.random.default_rng(2)
    W = np.abs(r.standard_normal((n, n))) / n
    x = torch.full((8, n), 15.0)
    recs = {}
    for which in ("reference", "amd"):
        for write in (False, True):
            net = ref(dt, dtype=torch.float32)
            if which == "reference":
                t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
                y0 = np.concatenate([np.full(n, -2.0), np.zeros(n)])
                args = (t(y0), torch.zeros(n), torch.zeros(n), t(W), t(np.ones(n)), t(np.ones(n)), t(np.full(n, -5.0)),
                        t(np.ones(n)))
                node = RefSpikeNet(make_qif(n), args, {"out": (n, 2 * n), "reset_var": (0, n), "qif_op/v": (0, n)},
                                   {"in": 0, "spike_var": 1, "weights": 2}, dt=dt, dtype=torch.float32)
            else:
                node = rectipy_amd.SpikeResetNet.from_pyrates(
                    "neuron_model_templates.spiking_neurons.qif.qif", input_var="qif_op/I_ext", output_var="qif_op/s",
                    weights=W, source_var="qif_op/s", target_var="qif_op/s_in", spike_var="qif_op/spike",
                    reset_var="qif_op/v", node_vars={"all/qif_op/v": -2.0, "all/qif_op/eta": -5.0}, dt=dt,
                    dtype=torch.float32, device="cpu", backend=oracle_backend)
            net.add_node("qif", node, node_type="diff_eq", op="qif_op")
            net.compile()
            for t_ in range(4):
                net.forward(x[t_])
            if write:
                with torch.no_grad():
                    net.state["qif"][:] = 1.0
            for t_ in range(4, 8):
                net.forward(x[t_])
            recs[which, write] = net.state["qif"].detach().numpy().copy()
    assert np.array_equal(recs["reference", False], recs["reference", True])
    assert np.array_equal(recs["amd", False], recs["amd", True])
    close(recs["amd", False], recs["reference", False], 1e-5)


def test_reference_fit_bptt_with_amd_node(ref, oracle_backend):
    """G5 through rectipy.Network.fit_bptt (the reference's window loop, loss, optimizer and detach calls) with the
    compiled node doing forward AND backward one step at a time."""
    g = gold("g5_bptt_one_window")
    net = ref(float(g["dt"]), dtype=torch.float32)
    node = amd_node(g, oracle_backend, train=["weights", "li_op/tau", "li_op/k", "li_op/eta"])
    net.add_node("rnn", node, node_type="diff_eq")
    net.add_func_node("out", g["W_out"].shape[0], "identity")
    edge = net.add_edge("rnn", "out", weights=g["W_out"].copy(), train="gd")
    lr = float(g["lr"])
    obs = net.fit_bptt(g["x"], g["target"], optimizer="sgd", lr=lr, update_steps=int(g["window"]), sampling_steps=1,
                       verbose=False)
    close(obs.to_numpy("out"), g["pred_f64"], 5e-5)
    for key, val, before in (("W", node["weights"], g["W"]), ("tau", node["li_op/tau"], g["tau"]),
                             ("k", node["li_op/k"], g["k"]), ("eta", node["li_op/eta"], g["eta"]),
                             ("W_out", edge.weights, g["W_out"])):
        ref_grad = (before - g[f"{key}_after_f64"]) / lr
        got_grad = (before - val.detach().cpu().numpy().astype(np.float64)) / lr
        floor = np.abs(before).max() * 1.2e-7 / lr
        assert np.abs(got_grad - ref_grad).max() <= 2e-3 * np.abs(ref_grad).max() + floor, key
