"""G32: surrogate-gradient BPTT through the reference's SpikeResetNet (rectipy/nodes.py:333-401, Spike.backward 479-481)
driven by the reference's Network.fit_bptt, one SGD window, at N = 512 and N = 1028 -- sizes where the production
branches of the spiking adjoint run (column-strip W^T u over row chunks, or the transposed copy; full 256-column chunks
plus a ragged one at 1028).  Gradients are read from the reference's `.grad`; of W's gradient the fixture keeps sampled
rows.  Inputs come from tests/golden/recipes.py (fingerprints stored).
Run in the build container only:  python tests/golden/make_golden_qif_bptt_large.py"""
import numpy as np
import torch

import recipes
from make_golden import import_reference, make_qif, np_, save


def network(n, seed):
    """QIF population with spread parameters; initial potentials spread so that many neurons spike inside the window."""
    W = (recipes.uniform((n, n), seed) * np.float32(20.0 / n)).astype(np.float32)
    eta = (-5.0 + recipes.normal(n, seed + 1)).astype(np.float32)
    tau = (0.8 + 0.4 * recipes.uniform(n, seed + 2)).astype(np.float32)
    k = (0.8 + 0.4 * recipes.uniform(n, seed + 3)).astype(np.float32)
    tau_s = (0.5 + recipes.uniform(n, seed + 4)).astype(np.float32)
    y0 = np.concatenate([(-5.0 + 65.0 * recipes.uniform(n, seed + 5)).astype(np.float32), recipes.uniform(n, seed + 6)])
    return dict(W=W, eta=eta, tau=tau, k=k, tau_s=tau_s, y0=y0)


def main():
    Network, RateNet, SpikeResetNet, Spike = import_reference()
    dt, kout, lr = 1e-3, 3, 1e-3
    for n, T in ((512, 120), (1028, 100)):
        P = network(n, 3200 + n)
        W_out = (recipes.normal((kout, n), 3250 + n) / np.float32(np.sqrt(n))).astype(np.float32)
        x = (15.0 + recipes.normal((T, n), 3260 + n)).astype(np.float32)
        target = (0.1 * recipes.normal((T, kout), 3270 + n)).astype(np.float32)
        rows = np.array(sorted(set(range(0, n, 37)) | {n - 1}))
        out = {}
        for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
            t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
            args = (t(P["y0"]), torch.zeros(n, dtype=dtype), torch.zeros(n, dtype=dtype), t(P["W"]), t(P["tau"]), t(P["k"]),
                    t(P["eta"]), t(P["tau_s"]))
            pm = {"in": 0, "spike_var": 1, "weights": 2, "qif_op/tau": 3, "qif_op/k": 4, "qif_op/eta": 5, "qif_op/tau_s": 6}
            vm = {"out": (n, 2 * n), "reset_var": (0, n)}
            node = SpikeResetNet(make_qif(n), args, vm, pm, spike_threshold=100.0, spike_reset=-100.0, dt=dt, dtype=dtype,
                                 device="cpu", train_params=["weights", "qif_op/tau", "qif_op/k", "qif_op/eta", "qif_op/tau_s"])
            net = Network(dt, dtype=dtype)
            net.add_node("rnn", node, node_type="diff_eq")
            net.add_func_node("out", kout, "identity")
            edge = net.add_edge("rnn", "out", weights=W_out.copy(), train="gd")
            obs = net.fit_bptt(x, target, optimizer="sgd", lr=lr, update_steps=T, sampling_steps=1, verbose=False)
            grads = {name: np_(node[key].grad) for name, key in (("W", "weights"), ("tau", "qif_op/tau"), ("k", "qif_op/k"),
                                                                ("eta", "qif_op/eta"), ("tau_s", "qif_op/tau_s"))}
            out["gW_rows_" + tag] = grads["W"][rows]
            out["gW_colsum_" + tag] = grads["W"].sum(axis=0)
            out["gW_fp_" + tag] = recipes.fingerprint(grads["W"])
            for name in ("tau", "k", "eta", "tau_s"):
                out[f"g{name}_{tag}"] = grads[name]
            out["gW_out_" + tag] = np_(edge.weights.grad)
            out["loss_" + tag] = float(obs["loss"][-1])
            out["pred_" + tag] = obs.to_numpy("out")
        scale = np.abs(out["gW_rows_f64"]).max()
        print(f"G32 n={n}: loss {out['loss_f64']:.6f}; reference f32-vs-f64: gW rows {np.abs(out['gW_rows_f64'] - out['gW_rows_f32']).max() / scale:.2e} "
              f"of max, geta {np.abs(out['geta_f64'] - out['geta_f32']).max() / np.abs(out['geta_f64']).max():.2e}, "
              f"pred {np.abs(out['pred_f64'] - out['pred_f32']).max():.2e}")
        save(f"g32_qif_bptt_n{n}", dt=dt, lr=lr, n=n, T=T, seed=3200 + n, rows=rows, theta=100.0, v_reset=-100.0, slope=0.5,
             fp_W=recipes.fingerprint(P["W"]), fp_x=recipes.fingerprint(x), fp_target=recipes.fingerprint(target),
             fp_y0=recipes.fingerprint(P["y0"]), **out)


if __name__ == "__main__":
    main()
