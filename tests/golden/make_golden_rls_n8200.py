"""G34: the reference's fit_rls at n_in = 8200 -- above the 8192 from which this package's RLS edge takes the SYMMETRIC pass
by default (only the blocks of P on and above the diagonal are streamed, row and column partials, mirror at the end) with the
one-workgroup-per-CU residency cap; 8 full 1024-blocks and a ragged ninth of 8 columns.  RLS.update rectipy/edges.py:227-234
driven by Network.fit_rls / _rls (rectipy/network.py:786-856, 1093-1121), float64; rnn (tanh, g = 0.8) -> rls edge -> output.

Run in the build container only:  python tests/golden/make_golden_rls_n8200.py   (about a minute)
Of P (8200 x 8200) the fixture keeps ten rows (as float32), the diagonal and a fingerprint of the float64 result."""
import numpy as np
import torch

import recipes
from make_golden import import_reference, np_, rate_node, save


def main():
    Network, RateNet, _, _ = import_reference()
    dt, kout, n, T, stride = 1e-2, 3, 8200, 12, 1
    P = recipes.li_network(n, seed=3400 + n, g=0.8, defaults=False)
    x = (0.5 * recipes.normal((T, n), seed=3450 + n)).astype(np.float32)
    target = recipes.normal((T, kout), seed=3460 + n)
    W0 = (0.1 * recipes.normal((kout, n), seed=3470 + n)).astype(np.float32)
    rows = np.array([0, 1023, 1024, 2500, 4095, 4096, 6000, 8191, 8192, 8199])  # block edges, both sides of the diagonal blocks
    dtype = torch.float64
    net = Network(dt, dtype=dtype)
    net.add_node("rnn", rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype), node_type="diff_eq")
    net.add_func_node("out", kout, "identity")
    edge = net.add_edge("rnn", "out", weights=W0.copy(), train="rls", beta=0.98, alpha=1.5, dtype=dtype)
    obs = net.fit_rls(x, target, update_steps=stride, sampling_steps=1, verbose=False)
    Pm = np_(edge.P)
    assert np.abs(Pm - Pm.T).max() < 1e-12 * np.abs(Pm).max()  # the reference's update keeps P symmetric
    save(f"g34_rls_n{n}", dt=dt, n=n, T=T, stride=stride, seed=3400 + n, beta=0.98, alpha=1.5, rows=rows,
         fp_W=recipes.fingerprint(P["W"]), fp_x=recipes.fingerprint(x), fp_target=recipes.fingerprint(target),
         fp_W0=recipes.fingerprint(W0), W_after_f64=np_(edge.weights), P_rows_f64=Pm[rows].astype(np.float32),
         P_diag_f64=np.diag(Pm).copy(), P_fp_f64=recipes.fingerprint(Pm), pred_f64=obs.to_numpy("out"),
         loss_f64=np.asarray([float(l) for l in obs["loss"]]), state_f64=np_(net.state["rnn"]))


if __name__ == "__main__":
    main()
