"""G14: recursive-least-squares readout training by the reference (RLS.update rectipy/edges.py:227-234 driven by
Network.fit_rls / _rls, rectipy/network.py:786-856, 1093-1121): rnn -> (rls edge) -> identity output.
Run in the build container only:  python tests/golden/make_golden_rls.py"""
import numpy as np
import torch

from make_golden import import_reference, np_, rate_node, rng_li, save


def main():
    Network, RateNet, _, _ = import_reference()
    n, kout, T, dt = 40, 3, 90, 1e-2
    P = rng_li(n, seed=141)
    r = np.random.default_rng(142)
    x = r.standard_normal((T, n)) * 0.5
    target = r.standard_normal((T, kout))
    W0 = r.standard_normal((kout, n)) * 0.1
    out = {}
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        for stride in (1, 4):
            net = Network(dt, dtype=dtype)
            net.add_node("rnn", rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype), node_type="diff_eq")
            net.add_func_node("out", kout, "identity")
            edge = net.add_edge("rnn", "out", weights=W0.copy(), train="rls", beta=0.95, alpha=2.0)
            obs = net.fit_rls(x, target, update_steps=stride, sampling_steps=1, verbose=False)
            key = f"s{stride}_{tag}"
            out["W_after_" + key] = np_(edge.weights)
            out["P_after_" + key] = np_(edge.P)
            out["pred_" + key] = obs.to_numpy("out")
            out["loss_" + key] = np.asarray([float(l) for l in obs["loss"]])
            out["state_" + key] = np_(net.state["rnn"])
    save("g14_rls", dt=dt, x=x, target=target, W0=W0, beta=0.95, alpha=2.0, **P, **out)


if __name__ == "__main__":
    main()
