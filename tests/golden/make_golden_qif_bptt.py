"""G10: surrogate-gradient BPTT through the reference's SpikeResetNet (rectipy/nodes.py:333-401, Spike.backward
479-481) driven by the reference's Network.fit_bptt -- one SGD window; parameter updates = -lr * gradients.
Run in the build container only:  python tests/golden/make_golden_qif_bptt.py
(see make_golden.py for how the reference is imported and why the field is hand-written)."""
import numpy as np
import torch

from make_golden import import_reference, make_qif, np_, save


def main():
    Network, RateNet, SpikeResetNet, Spike = import_reference()
    n, kout, T, dt = 24, 2, 160, 1e-3
    r = np.random.default_rng(101)
    W = r.uniform(0, 1, (n, n)) * 20.0 / n
    eta = -5.0 + r.standard_normal(n)
    tau, k, tau_s = r.uniform(0.8, 1.2, n), r.uniform(0.8, 1.2, n), r.uniform(0.5, 1.5, n)
    y0 = np.concatenate([r.uniform(-5.0, 60.0, n), r.uniform(0.0, 1.0, n)])  # several neurons spike inside the window
    W_out = r.standard_normal((kout, n)) / np.sqrt(n)
    x = 15.0 + r.standard_normal((T, n))
    target = r.standard_normal((T, kout)) * 0.1
    lr = 1e-3
    out = {}
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        args = (t(y0), torch.zeros(n, dtype=dtype), torch.zeros(n, dtype=dtype), t(W), t(tau), t(k), t(eta), t(tau_s))
        pm = {"in": 0, "spike_var": 1, "weights": 2, "qif_op/tau": 3, "qif_op/k": 4, "qif_op/eta": 5, "qif_op/tau_s": 6}
        vm = {"out": (n, 2 * n), "reset_var": (0, n)}
        node = SpikeResetNet(make_qif(n), args, vm, pm, spike_threshold=100.0, spike_reset=-100.0, dt=dt, dtype=dtype,
                             device="cpu", train_params=["weights", "qif_op/tau", "qif_op/k", "qif_op/eta",
                                                         "qif_op/tau_s"])
        net = Network(dt, dtype=dtype)
        net.add_node("rnn", node, node_type="diff_eq")
        net.add_func_node("out", kout, "identity")
        edge = net.add_edge("rnn", "out", weights=W_out.copy(), train="gd")
        obs = net.fit_bptt(x, target, optimizer="sgd", lr=lr, update_steps=T, sampling_steps=1, verbose=False)
        for name, key in (("W", "weights"), ("tau", "qif_op/tau"), ("k", "qif_op/k"), ("eta", "qif_op/eta"),
                          ("tau_s", "qif_op/tau_s")):
            out[f"{name}_after_{tag}"] = np_(node[key])
        out["W_out_after_" + tag] = np_(edge.weights)
        out["loss_" + tag] = np.asarray([float(l) for l in obs["loss"]])
        out["pred_" + tag] = obs.to_numpy("out")
        out["n_spikes_" + tag] = int(sum(int((obs_v >= 100.0).sum()) for obs_v in []))
    save("g10_qif_bptt", dt=dt, lr=lr, x=x, target=target, W=W, eta=eta, tau=tau, k=k, tau_s=tau_s, y0=y0, W_out=W_out,
         theta=100.0, v_reset=-100.0, slope=0.5, **out)


if __name__ == "__main__":
    main()
