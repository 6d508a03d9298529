"""G31: recursive-least-squares training by the REFERENCE at sizes where the production branches of the RLS kernels are
active (16-byte row pass with full 256-column chunks; the tiled pass with two full 1024-column tiles and a ragged third;
n_in % 4 != 0: the 4-byte pass): RLS.update rectipy/edges.py:227-234 driven by Network.fit_rls / _rls,
rectipy/network.py:786-856, 1093-1121.  rnn (tanh, contractive g = 0.8) -> (rls edge) -> identity output.

Run in the build container only:  python tests/golden/make_golden_rls_large.py
Large arrays are regenerated on both sides from tests/golden/recipes.py (fingerprints stored); of P (n x n) the fixture
keeps sampled rows, the diagonal and a fingerprint of the float64 result."""
import numpy as np
import torch

import recipes
from make_golden import import_reference, np_, rate_node, save


def main():
    Network, RateNet, _, _ = import_reference()
    dt, kout = 1e-2, 3
    for n, T, stride in ((520, 60, 1), (2060, 40, 1), (1030, 45, 3), (777, 40, 2)):
        P = recipes.li_network(n, seed=3100 + n, g=0.8, defaults=False)
        x = (0.5 * recipes.normal((T, n), seed=3150 + n)).astype(np.float32)
        target = recipes.normal((T, kout), seed=3160 + n)
        W0 = (0.1 * recipes.normal((kout, n), seed=3170 + n)).astype(np.float32)
        rows = np.array(sorted(set(range(0, n, 97)) | {n - 1}))
        out = {}
        for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
            net = Network(dt, dtype=dtype)
            net.add_node("rnn", rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype), node_type="diff_eq")
            net.add_func_node("out", kout, "identity")
            edge = net.add_edge("rnn", "out", weights=W0.copy(), train="rls", beta=0.98, alpha=1.5)
            obs = net.fit_rls(x, target, update_steps=stride, sampling_steps=1, verbose=False)
            Pm = np_(edge.P)
            out["W_after_" + tag] = np_(edge.weights)
            out["P_rows_" + tag] = Pm[rows]
            out["P_diag_" + tag] = np.diag(Pm).copy()
            out["P_fp_" + tag] = recipes.fingerprint(Pm)
            out["pred_" + tag] = obs.to_numpy("out")
            out["loss_" + tag] = np.asarray([float(l) for l in obs["loss"]])
            out["state_" + tag] = np_(net.state["rnn"])
        devW = np.abs(out["W_after_f64"] - out["W_after_f32"]).max()
        devP = np.abs(out["P_rows_f64"] - out["P_rows_f32"]).max() / np.abs(out["P_rows_f64"]).max()
        print(f"G31 n={n} stride={stride}: reference f32-vs-f64: W {devW:.3e} (|W| max {np.abs(out['W_after_f64']).max():.3f}), "
              f"P rel {devP:.3e}, pred {np.abs(out['pred_f64'] - out['pred_f32']).max():.3e}")
        save(f"g31_rls_n{n}", dt=dt, n=n, T=T, stride=stride, seed=3100 + n, beta=0.98, alpha=1.5, rows=rows,
             fp_W=recipes.fingerprint(P["W"]), fp_x=recipes.fingerprint(x), fp_target=recipes.fingerprint(target),
             fp_W0=recipes.fingerprint(W0), **out)


if __name__ == "__main__":
    main()
