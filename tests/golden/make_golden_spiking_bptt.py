"""G15-G17: surrogate-gradient BPTT through the reference's SpikeResetNet for the lif, qif_sfa and ik fields
(one SGD window of Network.fit_bptt; parameter updates = -lr * gradients).
Run in the build container only:  python tests/golden/make_golden_spiking_bptt.py"""
import numpy as np
import torch

from make_golden import import_reference, np_, save
from make_golden_spiking_models import field_ik, field_lif, field_qif_sfa


def run_case(Network, SpikeResetNet, name, field, n, y0, order, P, W, x, target, W_out, dt, theta, v_reset, lr, out_slice,
             extra=None):
    out = {}
    T = x.shape[0]
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        args = (t(y0), torch.zeros(x.shape[1], dtype=dtype), torch.zeros(n, dtype=dtype), t(W)) + tuple(t(P[q]) for q in order)
        pm = {"in": 0, "spike_var": 1, "weights": 2}
        pm.update({q: 3 + i for i, q in enumerate(order)})
        vm = {"out": out_slice, "reset_var": (0, n)}
        node = SpikeResetNet(field, args, vm, pm, spike_threshold=theta, spike_reset=v_reset, dt=dt, dtype=dtype,
                             device="cpu", train_params=["weights"] + list(order))
        net = Network(dt, dtype=dtype)
        net.add_node("rnn", node, node_type="diff_eq")
        net.add_func_node("out", W_out.shape[0], "identity")
        edge = net.add_edge("rnn", "out", weights=W_out.copy(), train="gd")
        obs = net.fit_bptt(x, target, optimizer="sgd", lr=lr, update_steps=T, sampling_steps=1, verbose=False)
        out["W_after_" + tag] = np_(node["weights"])
        for q in order:
            out[f"{q}_after_{tag}"] = np_(node[q])
        out["W_out_after_" + tag] = np_(edge.weights)
        out["loss_" + tag] = np.asarray([float(l) for l in obs["loss"]])
        out["pred_" + tag] = obs.to_numpy("out")
    save(name, dt=dt, lr=lr, x=x, target=target, W=W, y0=y0, W_out=W_out, theta=theta, v_reset=v_reset,
         slope=100.0 / abs(theta - v_reset), **P, **out, **(extra or {}))


def main():
    Network, _, SpikeResetNet, _ = import_reference()
    kout = 2

    # ---- G15 lif (input into I_ext) ------------------------------------------------------------------------
    n, T, dt = 24, 200, 1e-2
    r = np.random.default_rng(151)
    W = r.standard_normal((n, n)) * 2.0 / np.sqrt(n)
    P = dict(tau=r.uniform(5, 10, n), k=r.uniform(0.5, 1.5, n), eta=r.uniform(0.0, 0.5, n), tau_s=r.uniform(0.3, 0.7, n))
    y0 = np.concatenate([r.uniform(0, 0.95, n), r.uniform(0, 0.5, n)])
    x = 0.4 + 0.2 * r.standard_normal((T, n))
    run_case(Network, SpikeResetNet, "g15_lif_bptt", field_lif(n, False), n, y0, ("tau", "k", "eta", "tau_s"), P, W, x,
             r.standard_normal((T, kout)) * 0.1, r.standard_normal((kout, n)) / np.sqrt(n), dt, 1.0, 0.0, 1e-2, (n, 2 * n))

    # ---- G16 qif_sfa ---------------------------------------------------------------------------------------
    n, T, dt = 24, 160, 1e-3
    r = np.random.default_rng(161)
    W = r.uniform(0, 1, (n, n)) * 20.0 / n
    P = dict(tau=r.uniform(0.8, 1.2, n), k=r.uniform(0.8, 1.2, n), eta=-5.0 + r.standard_normal(n),
             tau_s=r.uniform(0.5, 1.5, n), alpha=r.uniform(0.5, 1.5, n), tau_x=r.uniform(5, 15, n))
    y0 = np.concatenate([r.uniform(-5.0, 60.0, n), r.uniform(0, 1, n), r.uniform(0, 0.5, n)])
    x = 15.0 + r.standard_normal((T, n))
    run_case(Network, SpikeResetNet, "g16_qif_sfa_bptt", field_qif_sfa(n), n, y0,
             ("tau", "k", "eta", "tau_s", "alpha", "tau_x"), P, W, x, r.standard_normal((T, kout)) * 0.1,
             r.standard_normal((kout, n)) / np.sqrt(n), dt, 100.0, -100.0, 1e-3, (n, 2 * n))

    # ---- G17 ik ----------------------------------------------------------------------------------------------
    n, T, dt = 24, 200, 1e-1
    r = np.random.default_rng(171)
    W = r.uniform(0, 1, (n, n)) * 15.0 / n
    P = dict(C=r.uniform(90, 110, n), k=r.uniform(0.6, 0.8, n), v_r=np.full(n, -60.0), v_theta=np.full(n, -40.0),  # noqa
             eta=r.standard_normal(n) * 5.0, g=r.uniform(0.8, 1.2, n), E_r=np.zeros(n), b=r.uniform(-2.5, -1.5, n),
             tau_u=r.uniform(30, 36, n), kappa=r.uniform(8, 12, n), tau_s=r.uniform(5, 7, n))
    order = ("C", "k", "v_r", "v_theta", "eta", "g", "E_r", "b", "tau_u", "kappa", "tau_s")
    y0 = np.concatenate([r.uniform(-60.0, 35.0, n), r.uniform(0, 20, n), r.uniform(0, 0.5, n)])
    x = 120.0 + 5.0 * r.standard_normal((T, n))
    # the field signature orders the parameters as `order`; every one of them is trained
    Network_, SpikeResetNet_ = Network, SpikeResetNet

    out = {}
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        args = (t(y0), torch.zeros(n, dtype=dtype), torch.zeros(n, dtype=dtype), t(W)) + tuple(t(P[q]) for q in order)
        pm = {"in": 0, "spike_var": 1, "weights": 2}
        pm.update({q: 3 + i for i, q in enumerate(order)})
        vm = {"out": (2 * n, 3 * n), "reset_var": (0, n)}
        train = ["weights"] + list(order)
        node = SpikeResetNet_(field_ik(n), args, vm, pm, spike_threshold=40.0, spike_reset=-55.0, dt=dt, dtype=dtype,
                              device="cpu", train_params=train)
        net = Network_(dt, dtype=dtype)
        net.add_node("rnn", node, node_type="diff_eq")
        net.add_func_node("out", kout, "identity")
        W_out = np.random.default_rng(172).standard_normal((kout, n)) / np.sqrt(n)
        target = np.random.default_rng(173).standard_normal((T, kout)) * 0.1
        edge = net.add_edge("rnn", "out", weights=W_out.copy(), train="gd")
        lr = 1e-3
        obs = net.fit_bptt(x, target, optimizer="sgd", lr=lr, update_steps=T, sampling_steps=1, verbose=False)
        out["W_after_" + tag] = np_(node["weights"])
        for q in train[1:]:
            out[f"{q}_after_{tag}"] = np_(node[q])
        out["W_out_after_" + tag] = np_(edge.weights)
        out["loss_" + tag] = np.asarray([float(l) for l in obs["loss"]])
        out["pred_" + tag] = obs.to_numpy("out")
    save("g17_ik_bptt", dt=dt, lr=lr, x=x, target=target, W=W, y0=y0, W_out=W_out, theta=40.0, v_reset=-55.0,
         slope=100.0 / 95.0, **P, **out)


if __name__ == "__main__":
    main()
