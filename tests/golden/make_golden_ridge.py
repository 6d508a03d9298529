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


def ill_conditioned():
    """G39: as many samples as neurons and a small ridge -- the solve amplifies how the states were rounded.  The reference's
    float64 run is the truth; its own float32 run shows what float32 STATES cost in this regime (the engine's states are
    float32 too; its Gram matrix and solve are float64)."""
    Network, RateNet, SpikeResetNet, Spike = import_reference()
    n, m, kout, dt, alpha = 64, 3, 2, 1e-2, 1e-3
    P = rng_li(n, seed=391)
    r = np.random.default_rng(392)
    W_in = r.standard_normal((n, m))
    out = {}
    for T, stride in ((60, 1), (600, 10), (120, 2)):      # T / stride = 60 samples ~ n = 64 unknowns per read-out channel
        x = r.standard_normal((T, m))
        mix = r.standard_normal((m, kout))
        target = np.tanh(np.cumsum(x, axis=0) * 0.05) @ mix + 0.01 * r.standard_normal((T, kout))
        key = f"T{T}_s{stride}"
        out["x_" + key], out["target_" + key] = x, target
        for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
            net = Network(dt, dtype=dtype)
            net.add_func_node("inp", m, "identity")
            net.add_node("rnn", rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype), node_type="diff_eq")
            net.add_edge("inp", "rnn", weights=W_in)
            # (the reference multiplies the sampled states with the UNsampled targets, network.py:765-768: sample the targets here)
            obs = net.fit_ridge(x, target[::stride] if stride > 1 else target, sampling_steps=stride, alpha=alpha, verbose=False) \
                if stride == 1 else None
            if obs is None:   # sampling_steps > 1: the reference's product fails on the shape mismatch; restate its solve on its states
                o = net.run(x, sampling_steps=stride, verbose=False, enable_grad=False)
                X = torch.stack(o["out"]) if isinstance(o["out"], list) else torch.as_tensor(o["out"])
                X = X.to(dtype)
                Y = torch.tensor(target[::stride], dtype=dtype)
                w = torch.inverse(X.T @ X + alpha * torch.eye(n, dtype=dtype)) @ X.T @ Y
                out[f"X_{key}_{tag}"], out[f"w_out_{key}_{tag}"] = np_(X), np_(w)
            else:
                out[f"X_{key}_{tag}"], out[f"w_out_{key}_{tag}"] = obs.to_numpy("out"), np_(obs["w_out"])
        G = out[f"X_{key}_f64"].astype(np.float64)
        s = np.linalg.svd(G, compute_uv=False)
        print(key, "samples", G.shape[0], "cond(X^T X + alpha I) = %.3g" % np.linalg.cond(G.T @ G + alpha * np.eye(n)),
              "smallest singular values", s[-3:],
              "w_out f32 vs f64: %.3g of %.3g" % (np.abs(out[f"w_out_{key}_f32"] - out[f"w_out_{key}_f64"]).max(),
                                                   np.abs(out[f"w_out_{key}_f64"]).max()))
    save("g39_fit_ridge_illcond", dt=dt, alpha=alpha, W_in=W_in, **P, **out)


if __name__ == "__main__":
    main()
    ill_conditioned()
