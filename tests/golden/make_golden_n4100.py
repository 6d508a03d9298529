"""G33: the reference's own classes at N = 4100 -- beyond the 4096 up to which the adjoint uses a transposed copy of W, so
the PRODUCTION adjoint (column strips over row chunks, the form N = 32768 runs) and a 33 x 33-tile dW GEMM with ragged
edge tiles meet reference-derived numbers; 16 full 256-column chunks + a ragged 4-column chunk in the forward row dot.

  run   Network.run, T = 60, sampling_steps = 1 (rectipy/network.py:542-601): every 8th column of the output, final state
  bptt  Network.fit_bptt, ONE SGD window of T = 40, all parameters trained (network.py:1016-1048, 1123-1130): gradients
        from the reference's float64 `.grad` -- 77 rows x 606 columns of dW, all row / column sums, the vectors

Run in the build container only:  python tests/golden/make_golden_n4100.py   (about a minute; see make_golden.py)
Large arrays are regenerated from tests/golden/recipes.py on both sides; the fixture keeps their fingerprints."""
import numpy as np
import torch

import recipes
from make_golden import import_reference, np_, rate_node, save
from make_golden_large import fp


def main():
    Network, RateNet, _, _ = import_reference()
    n, dt = 4100, 1e-2
    P = recipes.li_network(n, seed=3300 + n, g=1.5, defaults=False)
    out = {}
    # ---- run ----
    T = 60
    x = (0.5 * recipes.normal((T, n), seed=3310)).astype(np.float32)
    cols8 = np.arange(0, n, 8)
    for tag, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        net = Network(dt, dtype=dtype)
        net.add_node("rnn", rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype), node_type="diff_eq")
        obs = net.run(x, sampling_steps=1, verbose=False, enable_grad=False)
        out["run_out_" + tag] = obs.to_numpy("out")[:, cols8].astype(np.float64 if tag == "f64" else np.float32)
        out["run_state_" + tag] = np_(net.state["rnn"])
    out["run_ref_f32_dev"] = np.abs(out["run_out_f64"] - out["run_out_f32"]).max()
    print("run: reference f32-vs-f64 deviation", out["run_ref_f32_dev"])
    # ---- bptt ----
    Tb, kout, lr = 40, 3, 1e-2
    W_out = (recipes.normal((kout, n), seed=3320) / np.float32(np.sqrt(n))).astype(np.float32)
    xb = (0.5 * recipes.normal((Tb, n), seed=3330)).astype(np.float32)
    target = (0.1 * recipes.normal((Tb, kout), seed=3340)).astype(np.float32)
    rows = np.array(sorted(set(range(0, n, 64)) | set(range(n - 12, n))))
    cols = np.array(sorted(set(range(0, n, 7)) | set(range(n - 20, n))))
    dtype = torch.float64
    net = Network(dt, dtype=dtype)
    node = rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype,
                     train_params=["weights", "li_op/tau", "li_op/k", "li_op/eta"])
    net.add_node("rnn", node, node_type="diff_eq")
    net.add_func_node("out", kout, "identity")
    edge = net.add_edge("rnn", "out", weights=W_out.copy(), train="gd", dtype=dtype)
    obs = net.fit_bptt(xb, target, optimizer="sgd", lr=lr, update_steps=Tb, sampling_steps=1, verbose=False)
    gW = np_(node["weights"].grad)
    assert np.abs((P["W"].astype(np.float64) - np_(node["weights"])) / lr - gW).max() < 1e-9
    out.update(loss_f64=np.asarray([float(l) for l in obs["loss"]]), pred_f64=obs.to_numpy("out"), state_f64=np_(net.state["rnn"]),
               gW_sub=gW[np.ix_(rows, cols)].astype(np.float32), gW_rowsum=gW.sum(1), gW_colsum=gW.sum(0), gW_absmax=np.abs(gW).max(),
               g_tau=np_(node["li_op/tau"].grad), g_k=np_(node["li_op/k"].grad), g_eta=np_(node["li_op/eta"].grad),
               g_W_out=np_(edge.weights.grad), W_out_after=np_(edge.weights))
    save("g33_n4100", dt=dt, lr=lr, n=n, T=T, Tb=Tb, seed=3300 + n, rows=rows, cols=cols, cols8=cols8,
         **fp(W=P["W"], x=x, xb=xb, W_out=W_out, target=target), **out)


if __name__ == "__main__":
    main()
