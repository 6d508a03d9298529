"""G22: BPTT through the reference's RateNet with the sigmoid field (leaky_integrator.yaml:22,33): one SGD window of
Network.fit_bptt with EVERY parameter trained (W, tau, k, eta, r_max, s, v0, read-out); updates = -lr * gradient.
Run in the build container only:  python tests/golden/make_golden_sigmoid_bptt.py"""
import numpy as np
import torch

from make_golden import import_reference, li_sigmoid, np_, rng_li, save


def main():
    Network, RateNet, _, _ = import_reference()
    n, T, dt, kout, lr = 40, 120, 1e-2, 3, 1e-2
    P = rng_li(n, seed=221)
    r = np.random.default_rng(222)
    act = dict(r_max=r.uniform(0.5, 1.5, n), s=r.uniform(0.5, 2.5, n), v0=r.standard_normal(n) * 0.2)
    x = r.standard_normal((T, n)) * 0.5
    target = r.standard_normal((T, kout)) * 0.1
    W_out = r.standard_normal((kout, n)) / np.sqrt(n)
    names = ("tau", "k", "eta", "r_max", "s", "v0")
    out = {}
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        args = (t(P["y0"]), torch.zeros(n, dtype=dtype), t(P["W"])) + tuple(t({**P, **act}[q]) for q in names)
        pm = {"in": 0, "weights": 1}
        pm.update({q: 2 + i for i, q in enumerate(names)})
        node = RateNet(li_sigmoid, args, {"out": (0, n)}, pm, dt=dt, dtype=dtype, train_params=["weights"] + list(names))
        net = Network(dt, dtype=dtype)
        net.add_node("rnn", node, node_type="diff_eq")
        net.add_func_node("out", kout, "identity")
        edge = net.add_edge("rnn", "out", weights=W_out.copy(), train="gd")
        obs = net.fit_bptt(x, target, optimizer="sgd", lr=lr, update_steps=T, sampling_steps=1, verbose=False)
        out["W_after_" + tag] = np_(node["weights"])
        for q in names:
            out[f"{q}_after_{tag}"] = np_(node[q])
        out["W_out_after_" + tag] = np_(edge.weights)
        out["loss_" + tag] = np.asarray([float(l) for l in obs["loss"]])
        out["pred_" + tag] = obs.to_numpy("out")
    save("g22_sigmoid_bptt", dt=dt, lr=lr, x=x, target=target, W_out=W_out, **P, **act, **out)


if __name__ == "__main__":
    main()
