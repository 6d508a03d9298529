"""Edge variants and FeedbackNetwork against fixtures captured from the REFERENCE's own classes
(tests/golden/make_golden_topologies.py; reference rectipy/edges.py:68-174, rectipy/network.py:1196-1357).
On acyclic edges the masked / delay / filter variants are causal maps of the source's window, so such networks take the
fused-window plan (one library call per population and window); feedback graphs are walked step by step."""
import os

import numpy as np
import pytest
import torch

from rectipy_amd import FeedbackNetwork, Network, SpikeResetNet
from rectipy_amd.edge_variants import LinearFilter, LinearMasked, LinearMemory, LinearMemoryFilter

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
QIF = "neuron_model_templates.spiking_neurons.qif.qif"
TARGETS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


def gold(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


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


def li_node_kwargs(g, suffix=""):
    return dict(weights=g["W" + suffix], source_var="tanh_op/r", target_var="li_op/r_in", input_var="li_op/I_ext",
                output_var="li_op/v", dtype=torch.float32,
                node_vars={"all/li_op/tau": g["tau" + suffix], "all/li_op/k": g["k" + suffix],
                           "all/li_op/eta": g["eta" + suffix], "all/li_op/v": g["y0" + suffix]})


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5)])
def test_g27a_edge_classes_match_the_reference(dtype, tol):
    """The four edge classes driven directly: orientation rules (non-square (n_in, n_out) input AND square matrices are
    transposed, edges.py:22-26, 161-162), delay-buffer indexing with repeated delays (edges.py:91-94), filters."""
    g = gold("g27a_edge_variants")
    n_in, n_out = g["W"].shape[1], g["W"].shape[0]

    def seq(edge, data):
        return torch.stack([edge.forward(torch.tensor(row, dtype=dtype)) for row in data])

    cases = {
        "masked": (LinearMasked(n_in, n_out, mask=g["mask"], weights=g["W"], dtype=dtype), g["x"]),
        "masked_T": (LinearMasked(n_in, n_out, mask=np.ascontiguousarray(g["mask"].T), weights=g["Wt"], dtype=dtype), g["x"]),
        "masked_sq": (LinearMasked(5, 5, mask=g["mask_s"], weights=g["Ws"], dtype=dtype), g["xs"]),
        "memory": (LinearMemory(n_in, n_out, delays=g["delays"], weights=g["W"], dtype=dtype), g["x"]),
        "filter": (LinearFilter(n_in, n_out, filter_weights=g["F"], weights=g["W"], dtype=dtype), g["x"]),
        "memfilter": (LinearMemoryFilter(n_in, n_out, delays=g["delays"], filter_weights=g["F"], weights=g["W"],
                                         dtype=dtype), g["x"]),
    }
    for name, (edge, data) in cases.items():
        assert err(seq(edge, data), g[name + "_f64"]) <= tol, name
    with pytest.raises(ValueError):
        LinearFilter(n_in, n_out, filter_weights=np.zeros((3, 3)), weights=g["W"])
    with pytest.raises(ValueError):
        LinearMemory(n_in, n_out, delays=np.array([0, 1]), weights=g["W"])
    with pytest.raises(ValueError):
        LinearMasked(n_in, n_out, mask=np.zeros((3, 3)), weights=g["W"])


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g27b_edge_variants_inside_network_run(target):
    g = gold("g27b_edge_variants_in_network")
    n, m, k = g["W"].shape[0], g["W_in"].shape[1], g["W_out"].shape[0]
    net = Network(float(g["dt"]), **target)
    net.add_func_node("inp", m, "identity")
    net.add_diffeq_node("rnn", TANH, **li_node_kwargs(g))
    net.add_func_node("out", k, "identity")
    e1 = net.add_edge("inp", "rnn", weights=g["W_in"], mask=g["mask"])
    e2 = net.add_edge("rnn", "out", weights=g["W_out"], filter_weights=g["F"])
    assert isinstance(e1, LinearMasked) and isinstance(e2, LinearFilter)
    calls = count_window_calls(net)
    obs = net.run(g["x_m"], sampling_steps=1, verbose=False, enable_grad=False)
    assert err(obs.to_numpy("out"), g["masked_filter_f64"]) <= 2e-5
    assert net._chain_plan() is not None and calls == [len(g["x_m"])], "the fused plan: ONE window call for the run"
    for name, kw, cls in (("memory", dict(delays=g["delays"]), LinearMemory),
                          ("memfilter", dict(delays=g["delays"], filter_weights=g["F"]), LinearMemoryFilter)):
        net = Network(float(g["dt"]), **target)
        net.add_diffeq_node("rnn", TANH, **li_node_kwargs(g))
        net.add_func_node("out", k, "identity")
        assert type(net.add_edge("rnn", "out", weights=g["W_out"], **kw)) is cls
        calls = count_window_calls(net)
        obs = net.run(g["x_n"], sampling_steps=1, verbose=False, enable_grad=False)
        assert err(obs.to_numpy("out"), g[name + "_f64"]) <= 2e-5, name
        assert net._chain_plan() is not None and calls == [len(g["x_n"])], name


def count_window_calls(net):
    """Records the `steps` of every fused-window call the population receives (a step-by-step walk calls forward())."""
    node = net.get_node("rnn")
    calls, inner, step = [], node.run_window, node.forward

    def run_window(steps, *a, **kw):
        calls.append(steps)
        return inner(steps, *a, **kw)

    def forward(x):
        calls.append("step")
        return step(x)

    node.run_window, node.forward = run_window, forward
    return calls


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("sampling_steps,cutoff", [(1, 0), (3, 2)])
def test_windowed_edges_equal_the_step_by_step_walk(target, sampling_steps, cutoff):
    """inp -> [LinearMemoryFilter] -> rnn -> [LinearFilter] -> out, and a two-population chain joined by a delay edge:
    the fused-window plan (edges evaluated per window, state carried across the windows of two consecutive runs) gives
    what the reference's step-by-step graph walk gives, including the buffer-mean sampling (an edge with memory must see
    every step, so the read-out is applied per step and averaged afterwards)."""
    g = gold("g27b_edge_variants_in_network")
    n, m, k = g["W"].shape[0], g["W_in"].shape[1], g["W_out"].shape[0]
    rng = np.random.default_rng(3)
    delays_in = rng.integers(0, 4, m)
    F_in = rng.standard_normal((m, m)) * 0.3
    x = g["x_m"]

    def build(stepwise):
        net = Network(float(g["dt"]), **target)
        net.add_func_node("inp", m, "identity")
        net.add_diffeq_node("rnn", TANH, **li_node_kwargs(g))
        net.add_func_node("out", k, "identity")
        net.add_edge("inp", "rnn", weights=g["W_in"], delays=delays_in, filter_weights=F_in)
        net.add_edge("rnn", "out", weights=g["W_out"], filter_weights=g["F"])
        net._force_stepwise = stepwise
        return net

    outs = {}
    for stepwise in (False, True):
        net = build(stepwise)
        calls = count_window_calls(net)
        a = net.run(x[:17], sampling_steps=sampling_steps, cutoff=cutoff, verbose=False, enable_grad=False)
        b = net.run(x[17:], sampling_steps=sampling_steps, cutoff=cutoff, verbose=False, enable_grad=False)
        outs[stepwise] = (a.to_numpy("out"), b.to_numpy("out"), net.state["rnn"].cpu().numpy().copy())
        assert ("step" in calls) == stepwise
    for got, want in zip(outs[False], outs[True]):
        assert got.shape == want.shape and err(got, want) <= 2e-6

    # two populations joined by a delay edge (the multi-population plan)
    outs = {}
    W12 = rng.standard_normal((n, n)) * 0.2
    for stepwise in (False, True):
        net = Network(float(g["dt"]), **target)
        net.add_diffeq_node("rnn", TANH, **li_node_kwargs(g))
        kw2 = li_node_kwargs(g)
        kw2["output_var"] = "tanh_op/r"
        net.add_diffeq_node("rnn2", TANH, **kw2)
        net.add_edge("rnn", "rnn2", weights=W12, delays=np.arange(n) % 5)
        net._force_stepwise = stepwise
        obs = net.run(g["x_n"], sampling_steps=sampling_steps, cutoff=cutoff, verbose=False, enable_grad=False)
        outs[stepwise] = obs.to_numpy("out")
    assert err(outs[False], outs[True]) <= 2e-6


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g28_feedback_between_rate_populations(target):
    """(a) a -> b with feedback b -> a: the feedback edge reads b's CURRENT state (b is stepped after a);
    (c) a -> b -> c with a feedback edge a -> c: its source has already been stepped (network.py:1354-1357).
    Square edge matrices, i.e. the transposition rule is part of what is pinned."""
    g = gold("g28_feedback")
    net = FeedbackNetwork(float(g["dt"]), **target)
    net.add_diffeq_node("a", TANH, **li_node_kwargs(g, "_a"))
    net.add_diffeq_node("b", TANH, **li_node_kwargs(g, "_b"))
    net.add_edge("a", "b", weights=g["W_ab"])
    net.add_edge("b", "a", weights=g["W_ba"], feedback=True)
    net.compile()
    assert net._in_node == "a" and net._out_node == "b"
    obs = net.run(g["x"], sampling_steps=1, verbose=False, enable_grad=False)
    assert err(obs.to_numpy("out"), g["rate_ab_f64"]) <= 2e-5
    assert err(net.state["a"], g["rate_ab_state_a_f64"]) <= 2e-5 and err(net.state["b"], g["rate_ab_state_b_f64"]) <= 2e-5

    net = FeedbackNetwork(float(g["dt"]), **target)
    for lab in "abc":
        net.add_diffeq_node(lab, TANH, **li_node_kwargs(g, "_" + lab))
    net.add_edge("a", "b", weights=g["W_ab"])
    net.add_edge("b", "c", weights=g["W_bc"])
    net.add_edge("a", "c", weights=g["W_ac"], feedback=True)
    obs = net.run(g["x"], sampling_steps=1, verbose=False, enable_grad=False)
    assert err(obs.to_numpy("out"), g["rate_abc_f64"]) <= 2e-5


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_g28_feedback_between_spiking_populations(target):
    """QIF a -> QIF b with inhibitory feedback b -> a: a spiking node's `y` lags one step (nodes.py:387-392), so the
    feedback edge delivers b's previous output.  1500 steps with spikes in both populations."""
    g = gold("g28_feedback")
    nq, T = g["Wq0"].shape[0], int(g["Tq"])
    assert float(g["qif_ab_peak_f64"]) > 0.5, "the fixture must contain spikes"
    net = FeedbackNetwork(float(g["dtq"]), **target)
    for lab, i in (("a", 0), ("b", 1)):
        node = SpikeResetNet.from_pyrates(
            QIF, input_var="qif_op/I_ext", output_var="qif_op/s", weights=g[f"Wq{i}"], source_var="qif_op/s",
            target_var="qif_op/s_in", spike_var="qif_op/spike", reset_var="qif_op/v",
            node_vars={"all/qif_op/eta": g[f"etaq{i}"], "all/qif_op/v": g["y0q"][:nq], "all/qif_op/s": g["y0q"][nq:]},
            spike_threshold=100.0, spike_reset=-100.0, dt=float(g["dtq"]), dtype=torch.float32, **target)
        net.add_node(lab, node, node_type="diff_eq")
    net.add_edge("a", "b", weights=g["Wq_ab"])
    net.add_edge("b", "a", weights=g["Wq_ba"], feedback=True)
    x = np.full((T, nq), 15.0, dtype=np.float32)
    obs = net.run(x, sampling_steps=1, verbose=False, enable_grad=False)
    assert err(obs.to_numpy("out")[::5], g["qif_ab_f64"]) <= 5e-5


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_feedback_from_an_activation_node_and_fused_fallback(target):
    """Extension (the reference raises TypeError here: InstantNode["out"] is None): feedback from an activation node
    delivers its previous output.  Without feedback edges the class takes the fused path and agrees with Network."""
    rng = np.random.default_rng(2)
    n, k, T = 10, 2, 30
    W, W_out, W_fb = rng.standard_normal((n, n)) / 3, rng.standard_normal((k, n)), rng.standard_normal((n, k)) * 0.5
    net = FeedbackNetwork(1e-2, **target)
    net.add_diffeq_node("rnn", TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in",
                        input_var="li_op/I_ext", output_var="li_op/v", dtype=torch.float32)
    net.add_func_node("out", k, "tanh")
    net.add_edge("rnn", "out", weights=W_out)
    net.add_edge("out", "rnn", weights=W_fb, feedback=True)
    x = rng.standard_normal((T, n)).astype(np.float32)
    obs = net.run(x, sampling_steps=1, verbose=False, enable_grad=False)
    v, prev, ref = np.zeros(n), None, []
    for t in range(T):
        inp = x[t] + (W_fb @ prev if prev is not None else 0.0)
        out = np.tanh(W_out @ v)
        ref.append(out)
        v = v + 1e-2 * (-v / 10.0 + W @ np.tanh(v) + inp)
        prev = out
    assert err(obs.to_numpy("out"), np.asarray(ref)) < 1e-4
    net2 = FeedbackNetwork(1e-2, **target)
    net2.add_diffeq_node("rnn", TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in",
                         input_var="li_op/I_ext", output_var="li_op/v", dtype=torch.float32)
    net1 = Network(1e-2, **target)
    net1.add_diffeq_node("rnn", TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in",
                         input_var="li_op/I_ext", output_var="li_op/v", dtype=torch.float32)
    a = net2.run(x, verbose=False, enable_grad=False).to_numpy("out")
    b = net1.run(x, verbose=False, enable_grad=False).to_numpy("out")
    assert a.shape == (T, n) and np.array_equal(a, b)
