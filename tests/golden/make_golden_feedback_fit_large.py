"""G38: FeedbackNetwork.fit_bptt by the REFERENCE at sizes where the adjoint's 16-byte kernels are the code that runs
(rp_backward_coupled: chunk partials of W^T u and of the edges' W_e^T gx over full 256-column strips plus a ragged one,
populations of different widths): rate a (N = 260) -> rate b (N = 516) -> read-out (3) with feedback b -> a, ONE SGD window of
T = 40, every parameter trained.  W, the edges and the inputs are regenerated from tests/golden/recipes.py on both sides
(their fingerprints are stored); the fixture holds the reference's float64 gradients: the per-neuron vectors whole, of every
matrix its fingerprint, eight rows and eight columns.
Run in the build container only:  python tests/golden/make_golden_feedback_fit_large.py"""
import numpy as np
import torch

import recipes
from make_golden import import_reference, np_, rate_node, save


def edge(rows, cols, seed, g):
    return (recipes.normal((rows, cols), seed) * np.float32(g / np.sqrt(cols))).astype(np.float32)


def main():
    Network, RateNet, SpikeResetNet, Spike = import_reference()
    from rectipy.network import FeedbackNetwork
    na, nb, kout, T, dt, lr = 260, 516, 3, 40, 1e-2, 1e-2
    Pa = recipes.li_network(na, seed=3801, g=1.1, defaults=False)
    Pb = recipes.li_network(nb, seed=3802, g=1.1, defaults=False)
    W_ab, W_ba, W_out = edge(nb, na, 3803, 0.6), edge(na, nb, 3804, 0.6), edge(kout, nb, 3805, 1.0)
    x = (0.5 * recipes.normal((T, na), seed=3806)).astype(np.float32)
    target = (0.2 * recipes.normal((T, kout), seed=3807)).astype(np.float32)
    out = {"fp_" + name: recipes.fingerprint(arr) for name, arr in
           (("Wa", Pa["W"]), ("Wb", Pb["W"]), ("W_ab", W_ab), ("W_ba", W_ba), ("W_out", W_out), ("x", x), ("target", target))}
    names = ["weights", "li_op/tau", "li_op/k", "li_op/eta"]
    rows = {"a": np.array([0, 1, 63, 64, 128, 255, 256, 259]), "b": np.array([0, 5, 255, 256, 300, 511, 512, 515])}
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        net = FeedbackNetwork(dt)
        net.dtype = dtype
        nodes = {}
        for lab, P in (("a", Pa), ("b", Pb)):
            nodes[lab] = rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype, train_params=list(names))
            net.add_node(lab, nodes[lab], node_type="diff_eq")
        net.add_func_node("out", kout, "identity")
        e_ab = net.add_edge("a", "b", weights=W_ab.astype(np.float64), train="gd", dtype=dtype)
        e_out = net.add_edge("b", "out", weights=W_out.astype(np.float64), train="gd", dtype=dtype)
        e_ba = net.add_edge("b", "a", weights=W_ba.astype(np.float64), feedback=True, train="gd", dtype=dtype)
        obs = net.fit_bptt(torch.tensor(x, dtype=dtype), torch.tensor(target, dtype=dtype), optimizer="sgd", lr=lr,
                           update_steps=T, sampling_steps=1, verbose=False)
        for lab in "ab":
            for key in names[1:]:
                out[f"grad_{lab}_{key.split('/')[-1]}_{tag}"] = np_(nodes[lab][key].grad)
        mats = {"Wa": (nodes["a"]["weights"], "a", "a"), "Wb": (nodes["b"]["weights"], "b", "b"), "W_ab": (e_ab.weights, "b", "a"),
                "W_ba": (e_ba.weights, "a", "b"), "W_out": (e_out.weights, None, "b")}
        for key, (param, rlab, clab) in mats.items():
            g = np_(param.grad)
            out[f"grad_{key}_fp_{tag}"] = recipes.fingerprint(g)
            out[f"grad_{key}_rows_{tag}"] = g[rows[rlab]] if rlab else g
            out[f"grad_{key}_cols_{tag}"] = g[:, rows[clab]]
        out["pred_" + tag] = obs.to_numpy("out")
        out["loss_" + tag] = np.asarray([float(l) for l in obs["loss"]])
    dev = max(np.abs(out[k] - out[k[:-3] + "f64"]).max() / max(np.abs(out[k[:-3] + "f64"]).max(), 1e-30)
              for k in out if k.endswith("f32") and k.startswith("grad") and "_fp_" not in k)
    print("own float32 deviation of the reference's gradients (of each one's max):", dev)
    save("g38_feedback_fit_large", dt=dt, lr=lr, T=T, na=na, nb=nb, rows_a=rows["a"], rows_b=rows["b"], **out)


if __name__ == "__main__":
    main()
