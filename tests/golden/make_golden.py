"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own classes.

Run in the build container only (the reference lives at /root/reference and never travels):

    python tests/golden/make_golden.py

What is executed is the reference's `RateNet`, `SpikeResetNet`, `Spike`, `Network`
(`add_node`, `add_func_node`, `add_edge`, `compile`, `run`, `fit_bptt`), `Linear` and
`Observer` -- imported from /root/reference.  Two of the reference's imports are absent
from this image (`pyrates`, the code generator, and `multipledispatch`); they are
satisfied by inert stand-ins: a names-only `pyrates` module (none of its names is ever
called on the paths used here) and sympy's vendored copy of multipledispatch.  Because
the PyRates code generator is unavailable, the vector fields are written by hand from
the YAML equations, exactly as the reference's own test does
(rectipy_tests/test_nodes.py:32-33):
    leaky_integrator.yaml:10   v' = -v/tau + k*r_in + I_ext + eta ; :33 r = tanh(v);
                               :22 r = r_max/(1+exp(s*(v0-v)))
    qif.yaml:11-12             v' = (v^2+eta+I_ext)/tau + k*s_in ; s' = -s/tau_s + spike
with r_in = weights @ r (rows = targets, documentation/interfaces/model_definition.py:57).

Every fixture is stored in float64 ("truth") and float32 (what the fp32 engine is compared
with at a stated tolerance).  Only data is stored: inputs and expected outputs.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    import sympy.multipledispatch as md
    pr = types.ModuleType("pyrates")
    for name in ("NodeTemplate", "CircuitTemplate"):
        setattr(pr, name, type(name, (), {}))
    pr.clear = lambda *a, **k: None
    pr.clear_frontend_caches = lambda *a, **k: None
    sys.modules.setdefault("pyrates", pr)
    sys.modules.setdefault("multipledispatch", md)
    sys.path.insert(0, "/root/reference")
    import rectipy  # noqa: F401
    from rectipy import Network
    from rectipy.nodes import RateNet, Spike, SpikeResetNet
    return Network, RateNet, SpikeResetNet, Spike


# ---- hand-written vector fields (from the YAML equations) -----------------------------

def li_tanh(t, y, I_ext, weights, tau, k, eta):
    return -y / tau + k * (weights @ torch.tanh(y)) + I_ext + eta


def li_sigmoid(t, y, I_ext, weights, tau, k, eta, r_max, s, v0):
    r = r_max / (1 + torch.exp(s * (v0 - y)))
    return -y / tau + k * (weights @ r) + I_ext + eta


def make_qif(n):
    def qif(t, y, I_ext, spike, weights, tau, k, eta, tau_s):
        v, s = y[:n], y[n:]
        dv = (v ** 2 + eta + I_ext) / tau + k * (weights @ s)
        ds = -s / tau_s + spike
        return torch.cat((dv, ds), 0)
    return qif


LI_PARAM_MAP = {"in": 0, "weights": 1, "li_op/tau": 2, "li_op/k": 3, "li_op/eta": 4}


def rate_node(RateNet, W, tau, k, eta, y0, dt, dtype, train_params=None, n_in=None):
    n = W.shape[0]
    t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
    args = (t(y0), torch.zeros(n if n_in is None else n_in, dtype=dtype), t(W), t(tau), t(k), t(eta))
    return RateNet(li_tanh, args, {"out": (0, n), "li_op/v": (0, n)}, dict(LI_PARAM_MAP), dt=dt, dtype=dtype,
                   train_params=train_params)


def rng_li(n, seed, g=1.5):
    r = np.random.default_rng(seed)
    return dict(W=r.standard_normal((n, n)) * g / np.sqrt(n), tau=r.uniform(5, 10, n), k=r.uniform(0.5, 1.5, n),
                eta=r.standard_normal(n) * 0.1, y0=r.standard_normal(n))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", path, {k: np.asarray(v).shape for k, v in arrays.items()})


def np_(t):
    return t.detach().cpu().numpy()


def main():
    Network, RateNet, SpikeResetNet, Spike = import_reference()
    dtypes = {"f64": torch.float64, "f32": torch.float32}

    # ---- G1: RateNet.forward loop, full and broadcast input ---------------------------
    for n in (16, 64):
        P = rng_li(n, seed=100 + n)
        T, dt = 50, 1e-2
        r = np.random.default_rng(7)
        x_full = r.standard_normal((T, n)) * 0.5
        x_b = r.standard_normal((T, 1))
        out = {}
        for tag, dtype in dtypes.items():
            for xname, x in (("full", x_full), ("bcast", x_b)):
                node = rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype,
                                 n_in=x.shape[1])
                xs = torch.tensor(x, dtype=dtype)
                with torch.no_grad():
                    outs = np.stack([np_(node.forward(xs[t])) for t in range(T)])
                out[f"out_{xname}_{tag}"] = outs
                out[f"yT_{xname}_{tag}"] = np_(node.y)
        save(f"g1_rate_tanh_n{n}", dt=dt, x_full=x_full, x_bcast=x_b, **P, **out)

    # ---- G2: Network.run sampling / cutoff / record_vars semantics --------------------
    n, T, dt = 24, 40, 1e-2
    P = rng_li(n, seed=21)
    x = np.random.default_rng(22).standard_normal((T, n)) * 0.5
    out = {}
    for tag, dtype in dtypes.items():
        for s in (1, 2, 3):
            for reduce in (False, True):
                net = Network(dt, dtype=dtype)
                node = rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype)
                net.add_node("rnn", node, node_type="diff_eq")
                obs = net.run(torch.tensor(x, dtype=dtype), sampling_steps=s, cutoff=5, verbose=False,
                              enable_grad=False, record_vars=[("rnn", "li_op/v", reduce)])
                key = f"s{s}_{'red' if reduce else 'full'}_{tag}"
                out["out_" + key] = obs.to_numpy("out")
                out["steps_" + key] = np.asarray(obs["steps"])
                out["v_" + key] = obs.to_numpy(("rnn", "li_op/v"))
                out["state_" + key] = np_(net.state["rnn"])
    save("g2_run_sampling", dt=dt, x=x, cutoff=5, **P, **out)

    # ---- G3: inp -> rnn -> out with Linear edges ---------------------------------------
    n, m, kout, T, dt = 32, 3, 2, 60, 1e-2
    P = rng_li(n, seed=31)
    r = np.random.default_rng(32)
    W_in, W_out = r.standard_normal((n, m)), r.standard_normal((kout, n)) / np.sqrt(n)
    x = r.standard_normal((T, m))
    out = {}
    for tag, dtype in dtypes.items():
        net = Network(dt, dtype=dtype)
        net.add_func_node("inp", m, "identity")
        net.add_node("rnn", rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype),
                     node_type="diff_eq")
        net.add_func_node("out", kout, "identity")
        net.add_edge("inp", "rnn", weights=W_in)
        net.add_edge("rnn", "out", weights=W_out)
        obs = net.run(x, sampling_steps=2, verbose=False, enable_grad=False)
        out["out_" + tag] = obs.to_numpy("out")
        out["steps_" + tag] = np.asarray(obs["steps"])
        out["state_" + tag] = np_(net.state["rnn"])
    save("g3_in_rnn_out", dt=dt, x=x, W_in=W_in, W_out=W_out, **P, **out)

    # ---- G4: SpikeResetNet with the QIF field ------------------------------------------
    n, T, dt = 32, 3000, 1e-3
    r = np.random.default_rng(41)
    W = r.uniform(0, 1, (n, n)) * 20.0 / n
    eta = -5.0 + r.standard_normal(n)
    tau, k, tau_s = np.ones(n), np.ones(n), np.ones(n)
    y0 = np.concatenate([np.full(n, -2.0), np.zeros(n)])
    out = {}
    for tag, dtype in dtypes.items():
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        args = (t(y0), torch.zeros(1, dtype=dtype), torch.zeros(n, dtype=dtype), t(W), t(tau), t(k), t(eta), t(tau_s))
        pm = {"in": 0, "spike_var": 1, "weights": 2, "qif_op/tau": 3, "qif_op/k": 4, "qif_op/eta": 5, "qif_op/tau_s": 6}
        vm = {"out": (n, 2 * n), "reset_var": (0, n), "qif_op/v": (0, n), "qif_op/s": (n, 2 * n)}
        node = SpikeResetNet(make_qif(n), args, vm, pm, spike_threshold=100.0, spike_reset=-100.0, dt=dt, dtype=dtype,
                             device="cpu")
        xin = torch.tensor([15.0], dtype=dtype)
        outs, vs, events = [], [], []
        with torch.no_grad():
            for step in range(T):
                v_before = np_(node._y_spike).copy()
                outs.append(np_(node.forward(xin)).copy())
                vs.append(v_before)
                for idx in np.nonzero(v_before >= 100.0)[0]:
                    events.append((step, idx))
        out["s_" + tag] = np.stack(outs)[::10]
        out["v_" + tag] = np.stack(vs)[::10]
        out["events_" + tag] = np.asarray(events, dtype=np.int32).reshape(-1, 2)
        out["v_final_" + tag] = np_(node._y_spike)
        out["s_final_" + tag] = np_(node._y_stop)
    save("g4_qif_spikes", dt=dt, W=W, eta=eta, tau=tau, k=k, tau_s=tau_s, y0=y0, x_const=15.0, theta=100.0,
         v_reset=-100.0, stride=10, **out)

    # ---- G5 / G6: fit_bptt windows (SGD) -> parameter updates = -lr * gradients --------
    n, kout, dt = 48, 3, 1e-2
    P = rng_li(n, seed=51)
    r = np.random.default_rng(52)
    W_out = r.standard_normal((kout, n)) / np.sqrt(n)
    for name, T, win in (("g5_bptt_one_window", 100, 100), ("g6_bptt_two_windows", 120, 60)):
        x = r.standard_normal((T, n)) * 0.5
        target = r.standard_normal((T, kout)) * 0.1
        out = {}
        for tag, dtype in dtypes.items():
            net = Network(dt, dtype=dtype)
            node = rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype,
                             train_params=["weights", "li_op/tau", "li_op/k", "li_op/eta"])
            net.add_node("rnn", node, node_type="diff_eq")
            net.add_func_node("out", kout, "identity")
            edge = net.add_edge("rnn", "out", weights=W_out.copy(), train="gd")
            lr = 1e-2
            obs = net.fit_bptt(x, target, optimizer="sgd", lr=lr, update_steps=win, sampling_steps=1, verbose=False)
            out["W_after_" + tag] = np_(node["weights"])
            out["tau_after_" + tag] = np_(node["li_op/tau"])
            out["k_after_" + tag] = np_(node["li_op/k"])
            out["eta_after_" + tag] = np_(node["li_op/eta"])
            out["W_out_after_" + tag] = np_(edge.weights)
            out["loss_" + tag] = np.asarray([float(l) for l in obs["loss"]])
            out["pred_" + tag] = obs.to_numpy("out")
            out["state_" + tag] = np_(net.state["rnn"])
        save(name, dt=dt, lr=lr, window=win, x=x, target=target, W_out=W_out, **P, **out)

    # ---- G7: Spike forward / surrogate backward on a grid incl. x == 0 -----------------
    grid = np.array([-3.0, -1.0, -1e-3, 0.0, 1e-3, 0.5, 2.0, 50.0])
    out = {}
    for tag, dtype in dtypes.items():
        Spike.center = torch.tensor(1.0, dtype=dtype)
        Spike.slope = torch.tensor(100.0 / 200.0, dtype=dtype)
        xg = torch.tensor(grid, dtype=dtype, requires_grad=True)
        y = Spike.apply(xg)
        y.backward(torch.ones_like(y))
        out["fwd_" + tag], out["bwd_" + tag] = np_(y), np_(xg.grad)
    save("g7_spike_fn", grid=grid, slope=0.5, **out)

    # ---- G8: sigmoid field ---------------------------------------------------------------
    n, T, dt = 32, 50, 1e-2
    P = rng_li(n, seed=81)
    r = np.random.default_rng(82)
    r_max, s, v0 = r.uniform(0.5, 1.5, n), r.uniform(0.5, 2.5, n), r.standard_normal(n) * 0.2
    x = r.standard_normal((T, n)) * 0.5
    out = {}
    for tag, dtype in dtypes.items():
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        args = (t(P["y0"]), torch.zeros(n, dtype=dtype), t(P["W"]), t(P["tau"]), t(P["k"]), t(P["eta"]), t(r_max), t(s), t(v0))
        node = RateNet(li_sigmoid, args, {"out": (0, n)}, {"in": 0, "weights": 1}, dt=dt, dtype=dtype)
        xs = torch.tensor(x, dtype=dtype)
        with torch.no_grad():
            out["out_" + tag] = np.stack([np_(node.forward(xs[i])) for i in range(T)])
        out["yT_" + tag] = np_(node.y)
    save("g8_rate_sigmoid", dt=dt, x=x, r_max=r_max, s=s, v0=v0, **P, **out)

    # ---- G9: contractive long run (g = 0.5) ---------------------------------------------
    n, T, dt = 64, 2000, 1e-2
    P = rng_li(n, seed=91, g=0.5)
    x = np.random.default_rng(92).standard_normal((T, 1))
    out = {}
    for tag, dtype in dtypes.items():
        net = Network(dt, dtype=dtype)
        net.add_node("rnn", rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype, n_in=1),
                     node_type="diff_eq")
        obs = net.run(torch.tensor(x, dtype=dtype), sampling_steps=20, verbose=False, enable_grad=False)
        out["out_" + tag] = obs.to_numpy("out")
        out["state_" + tag] = np_(net.state["rnn"])
    save("g9_contractive_long", dt=dt, x=x, **P, **out)


if __name__ == "__main__":
    main()
