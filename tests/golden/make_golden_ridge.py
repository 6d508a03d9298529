"""G26: the reference's Network.fit_ridge (rectipy/network.py:709-784, closed-form solve 765-768) followed by
Network.test (898-944) through the read-out it added -- inp -> Linear -> tanh population (N = 64), float64 and float32.
Run in the build container only:  python tests/golden/make_golden_ridge.py
(see make_golden.py for how the reference is imported and why the field is hand-written)."""
import numpy as np
import torch

from make_golden import import_reference, np_, rate_node, rng_li, save


def main():
    Network, RateNet, SpikeResetNet, Spike = import_reference()
    n, m, kout, T, T2, dt, alpha = 64, 3, 2, 400, 150, 1e-2, 1e-1
    P = rng_li(n, seed=261)
    r = np.random.default_rng(262)
    W_in = r.standard_normal((n, m))
    x, x2 = r.standard_normal((T, m)), r.standard_normal((T2, m))
    mix = r.standard_normal((m, kout))
    target = np.tanh(np.cumsum(x, axis=0) * 0.05) @ mix + 0.01 * r.standard_normal((T, kout))
    target2 = np.tanh(np.cumsum(x2, axis=0) * 0.05) @ mix
    out = {}
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        net = Network(dt, dtype=dtype)
        net.add_func_node("inp", m, "identity")
        net.add_node("rnn", rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype),
                     node_type="diff_eq")
        net.add_edge("inp", "rnn", weights=W_in)
        obs = net.fit_ridge(x, target, sampling_steps=1, alpha=alpha, verbose=False)
        out["X_" + tag] = obs.to_numpy("out")
        out["w_out_" + tag] = np_(obs["w_out"])
        out["y_" + tag] = np_(obs["y"])
        out["state_fit_" + tag] = np_(net.state["rnn"])
        assert "readout" in net.nodes
        obs2, loss = net.test(x2, target2, loss="mse", sampling_steps=1, verbose=False)
        out["test_out_" + tag] = obs2.to_numpy("out")
        out["test_loss_" + tag] = loss
        out["state_test_" + tag] = np_(net.state["rnn"])
        G = out["X_" + tag].astype(np.float64)
        print(tag, "cond(X^T X + alpha I) =", np.linalg.cond(G.T @ G + alpha * np.eye(n)), "test loss", loss)
    print("w_out f32 vs f64:", np.abs(out["w_out_f32"] - out["w_out_f64"]).max(), "scale", np.abs(out["w_out_f64"]).max())
    print("y f32 vs f64:", np.abs(out["y_f32"] - out["y_f64"]).max())
    save("g26_fit_ridge", dt=dt, alpha=alpha, x=x, x2=x2, target=target, target2=target2, W_in=W_in, **P, **out)


if __name__ == "__main__":
    main()
