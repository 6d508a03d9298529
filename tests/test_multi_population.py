"""Acyclic networks of several populations (input -> p1 -> p2 [+ skip edge] -> read-out; a spiking layer behind a rate
layer): evaluated window by window -- one fused window per population -- with the same results as the step-by-step graph
walk of the reference (network.py:462-478, 586-599): outputs, buffer-mean sampling, recorded variables, gradients."""
import numpy as np
import pytest
import torch

from rectipy_amd import Network

TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
LIF = "neuron_model_templates.spiking_neurons.lif.lif"
TARGETS = [pytest.param("oracle", id="oracle"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture
def target(request, oracle_backend):
    if request.param == "oracle":
        return dict(device="cpu", backend=oracle_backend)
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return dict(device="cuda:0", backend=None)


def build(target, train=False, spiking_second=False, stepwise=False):
    rng = np.random.default_rng(3)
    n1, n2, m, k = 24, 16, 3, 2
    net = Network(1e-2, **target)
    net._force_stepwise = stepwise
    net.add_func_node("inp", m, "identity")
    net.add_diffeq_node("p1", TANH, weights=rng.standard_normal((n1, n1)) / 4, source_var="tanh_op/r",
                        target_var="li_op/r_in", input_var="li_op/I_ext", output_var="tanh_op/r", dtype=torch.float32,
                        train_params=["weights", "li_op/tau"] if train else None,
                        node_vars={"all/li_op/tau": rng.uniform(0.5, 1.5, n1)})
    if spiking_second:
        net.add_diffeq_node("p2", LIF, weights=np.abs(rng.standard_normal((n2, n2))) / 4, source_var="lif_op/s",
                            target_var="lif_op/s_in", input_var="lif_op/I_ext", output_var="lif_op/s",
                            spike_var="lif_op/spike", reset_var="lif_op/v", spike_threshold=1.0, spike_reset=0.0,
                            dtype=torch.float32, train_params=["weights"] if train else None,
                            node_vars={"all/lif_op/tau": 0.5, "all/lif_op/eta": 4.0, "all/lif_op/tau_s": 0.2})
    else:
        net.add_diffeq_node("p2", TANH, weights=rng.standard_normal((n2, n2)) / 4, source_var="tanh_op/r",
                            target_var="li_op/r_in", input_var="li_op/I_ext", output_var="li_op/v", dtype=torch.float32,
                            train_params=["weights"] if train else None, node_vars={"all/li_op/tau": 0.8})
    net.add_func_node("out", k, "tanh")
    net.add_edge("inp", "p1", weights=rng.standard_normal((n1, m)), train="gd" if train else None)
    net.add_edge("p1", "p2", weights=rng.standard_normal((n2, n1)) / 3, train="gd" if train else None)
    net.add_edge("inp", "p2", weights=rng.standard_normal((n2, m)) / 2)          # skip edge: p2 has two predecessors
    net.add_edge("p2", "out", weights=rng.standard_normal((k, n2)), train="gd" if train else None)
    return net


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("spiking_second", [False, True])
def test_run_window_by_window_equals_step_by_step(target, spiking_second):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((90, 3)).astype(np.float32)
    second_var = ("p2", "lif_op/v" if spiking_second else "li_op/v", False)
    res = []
    for stepwise in (False, True):
        net = build(target, spiking_second=spiking_second, stepwise=stepwise)
        net.compile()
        assert (net._dag_plan([]) is None) == stepwise
        obs = net.run(x, sampling_steps=4, cutoff=6, verbose=False, enable_grad=False,
                      record_vars=[("p1", "li_op/v", False), second_var])
        res.append((obs["steps"], obs.to_numpy("out"), obs.to_numpy(("p1", "li_op/v")), obs.to_numpy(second_var[:2]),
                    {k: v.detach().cpu().numpy() for k, v in net.state.items()}))
    assert list(res[0][0]) == list(res[1][0]) and len(res[0][0]) > 10
    for a, b in zip(res[0][1:4], res[1][1:4]):
        assert a.shape == b.shape and np.abs(a - b).max() < 2e-6
    for key in res[0][4]:
        assert np.abs(res[0][4][key] - res[1][4][key]).max() < 2e-6, key
    if spiking_second:
        assert np.abs(np.diff(res[0][3], axis=0)).max() > 0.3, "the LIF layer must spike"


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("spiking_second", [False, True])
def test_fit_bptt_window_by_window_equals_step_by_step(target, spiking_second):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((60, 3)).astype(np.float32)
    tgt = (rng.standard_normal((60, 2)) * 0.2).astype(np.float32)
    res = []
    for stepwise in (False, True):
        net = build(target, train=True, spiking_second=spiking_second, stepwise=stepwise)
        obs = net.fit_bptt(x, tgt, optimizer="sgd", lr=0.05, update_steps=20, sampling_steps=5, verbose=False)
        params = {"W1": net.get_node("p1")["weights"], "tau1": net.get_node("p1")["li_op/tau"],
                  "W2": net.get_node("p2")["weights"], "W_in": net.get_edge("inp", "p1").weights,
                  "W_12": net.get_edge("p1", "p2").weights, "W_out": net.get_edge("p2", "out").weights}
        res.append(({k: v.detach().cpu().numpy().copy() for k, v in params.items()}, obs.to_numpy("loss"),
                    obs.to_numpy("out")))
    for key in res[0][0]:
        a, b = res[0][0][key], res[1][0][key]
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(b).max()), key
    assert np.abs(res[0][1] - res[1][1]).max() < 1e-5 and res[0][1].max() > 0
    assert np.abs(res[0][2] - res[1][2]).max() < 1e-5


@pytest.mark.parametrize("target", TARGETS, indirect=True)
@pytest.mark.parametrize("window", [1, 2, 3])
def test_short_windows_leave_unreached_gradients_none(target, window):
    """Every population delays by one step, so in a window of 1 or 2 steps the upstream parameters cannot reach any
    prediction: the step-by-step autograd graph (the reference's) has no path to them and leaves ``grad`` None -- Adam
    then skips them -- while the fused windows' adjoints return zeros.  Network._unreached_parameters restores None.
    A gradient-tracking run first gives Adam non-zero moments in the first window (through the older graph), which is
    what makes zeros and None differ afterwards."""
    rng = np.random.default_rng(6)
    x0 = rng.standard_normal((5, 3)).astype(np.float32)
    x = rng.standard_normal((4 * window, 3)).astype(np.float32)
    tgt = (rng.standard_normal((4 * window, 2)) * 0.2).astype(np.float32)
    res = []
    for stepwise in (False, True):
        net = build(target, train=True, stepwise=stepwise)
        net.run(x0, sampling_steps=1, verbose=False, enable_grad=True)
        net.fit_bptt(x, tgt, optimizer="adam", lr=0.01, update_steps=window, sampling_steps=1, verbose=False)
        params = {"W1": net.get_node("p1")["weights"], "tau1": net.get_node("p1")["li_op/tau"],
                  "W2": net.get_node("p2")["weights"], "W_in": net.get_edge("inp", "p1").weights,
                  "W_12": net.get_edge("p1", "p2").weights, "W_out": net.get_edge("p2", "out").weights}
        res.append(({k: v.detach().cpu().numpy().copy() for k, v in params.items()},
                    {k for k, v in params.items() if v.grad is None}))
    # (the step-by-step arm walks the graph node by node as the reference does, with the same rule applied to it)
    expected = {1: {"W1", "tau1", "W_in", "W2", "W_12"}, 2: {"W1", "tau1", "W_in"}, 3: set()}[window]
    assert res[0][1] == expected and res[1][1] == expected
    for key in res[0][0]:
        a, b = res[0][0][key], res[1][0][key]
        assert np.abs(a - b).max() <= 2e-4 * max(1.0, np.abs(b).max()), key


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_fit_rls_behind_two_populations(target):
    """RLS read-out behind inp -> p1 -> p2: the populations' windows first, then one RLS sweep; same weights, P matrix
    and predictions as the step-by-step loop."""
    rng = np.random.default_rng(6)
    x = rng.standard_normal((80, 3)).astype(np.float32)
    tgt = (rng.standard_normal((80, 2)) * 0.3).astype(np.float32)
    res = []
    for stepwise in (False, True):
        net = build(target, stepwise=stepwise)
        net.pop_edge("p2", "out")
        net.pop_node("out")
        net.add_func_node("out", 2, "identity")
        edge = net.add_edge("p2", "out", train="rls", beta=0.98, alpha=1.0)
        obs = net.fit_rls(x, tgt, update_steps=2, sampling_steps=1, verbose=False)
        res.append((edge.weights.detach().cpu().numpy().copy(), edge.P.detach().cpu().numpy().copy(),
                    obs.to_numpy("out"), obs.to_numpy("loss")))
    for a, b in zip(res[0], res[1]):
        assert a.shape == b.shape and np.abs(a - b).max() <= 2e-4 * max(1.0, np.abs(b).max())
    assert np.abs(res[0][0]).max() > 0


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_softmax_read_out_in_window_mode(target):
    """A normalising read-out (softmax over the output units, the reference builds it with dim=0 for one-step vectors)
    behind a population: applied per step inside the fused window evaluation; buffer means, training and outputs equal
    the step-by-step walk."""
    rng = np.random.default_rng(7)
    n, k, T = 20, 4, 50
    W, W_out = rng.standard_normal((n, n)) / 4, rng.standard_normal((k, n))
    x = rng.standard_normal((T, 1)).astype(np.float32)
    tgt = np.eye(k, dtype=np.float32)[rng.integers(0, k, T)]
    res = []
    for stepwise in (False, True):
        net = Network(1e-2, **target)
        net._force_stepwise = stepwise
        net.add_diffeq_node("rnn", TANH, weights=W, source_var="tanh_op/r", target_var="li_op/r_in",
                            input_var="li_op/I_ext", output_var="tanh_op/r", dtype=torch.float32,
                            node_vars={"all/li_op/tau": 0.5})
        net.add_func_node("out", k, "softmax")
        edge = net.add_edge("rnn", "out", weights=W_out.copy(), train="gd")
        obs = net.run(x, sampling_steps=3, cutoff=4, verbose=False, enable_grad=False)
        out = obs.to_numpy("out")
        assert np.abs(out.sum(axis=1) - 1.0).max() < 1e-5          # a mean of probability vectors
        net.reset()
        net.fit_bptt(x, tgt, optimizer="sgd", lr=0.5, update_steps=25, sampling_steps=5, verbose=False)
        res.append((out, edge.weights.detach().cpu().numpy().copy()))
    for a, b in zip(res[0], res[1]):
        assert np.abs(a - b).max() < 1e-5
    assert np.abs(res[0][1] - W_out).max() > 1e-4


@pytest.mark.parametrize("target", TARGETS, indirect=True)
def test_epoch_training_of_two_populations(target):
    """fit_bptt with a LIST of inputs (one optimisation step per epoch, state restored in between, reference
    network.py:983-1014) on the two-population network: window evaluation == step-by-step walk."""
    rng = np.random.default_rng(8)
    xs = [rng.standard_normal((30, 3)).astype(np.float32) for _ in range(3)]
    ts = [(rng.standard_normal((30, 2)) * 0.2).astype(np.float32) for _ in range(3)]
    res = []
    for stepwise in (False, True):
        net = build(target, train=True, stepwise=stepwise)
        obs = net.fit_bptt(xs, ts, optimizer="sgd", lr=0.05, sampling_steps=1, verbose=False)
        res.append((net.get_node("p1")["weights"].detach().cpu().numpy().copy(),
                    net.get_edge("p1", "p2").weights.detach().cpu().numpy().copy(), np.asarray(obs["epoch_loss"])))
    for a, b in zip(res[0], res[1]):
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(b).max())
    assert len(res[0][2]) == 3 and res[0][2].max() > 0
