"""G37: FeedbackNetwork.fit_bptt by the REFERENCE (rectipy/network.py:1196-1357 evaluated by the training loop 1016-1048):
SGD windows through a cycle, every parameter trained (updates = -lr * gradient; in (a) the second window starts from the
updated parameters and the detached state):
  (a) rate a (N = 40) -> rate b (N = 56) -> identity read-out (3), feedback b -> a; T = 60, windows of 30;
  (b) QIF a -> QIF b (N = 36 each) -> read-out (2), feedback b -> a; ONE window of T = 100 (the reference's detach leaves
      the spike argument of a SpikeResetNet attached, so its second backward fails -- as in a plain Network), several
      spikes in both populations (the feedback edge reads b's state of the PREVIOUS step, nodes.py:387-392, and the
      gradient passes back through that read).
Run in the build container only:  python tests/golden/make_golden_feedback_fit.py"""
import numpy as np
import torch

from make_golden import import_reference, make_qif, np_, rate_node, rng_li, save


def main():
    Network, RateNet, SpikeResetNet, Spike = import_reference()
    from rectipy.network import FeedbackNetwork
    dtypes = {"f64": torch.float64, "f32": torch.float32}
    out = {}
    # ---- (a) rate populations -----------------------------------------------------------------------------------
    na, nb, kout, T, win, dt, lr = 40, 56, 3, 60, 30, 1e-2, 5e-2
    Pa, Pb = rng_li(na, seed=3701, g=1.2), rng_li(nb, seed=3702, g=1.2)
    r = np.random.default_rng(3703)
    W_ab, W_ba = r.standard_normal((nb, na)) * 0.6 / np.sqrt(na), r.standard_normal((na, nb)) * 0.6 / np.sqrt(nb)
    W_out = r.standard_normal((kout, nb)) / np.sqrt(nb)
    x, target = r.standard_normal((T, na)) * 0.5, r.standard_normal((T, kout)) * 0.2
    names = ["weights", "li_op/tau", "li_op/k", "li_op/eta"]
    for tag, dtype in dtypes.items():
        net = FeedbackNetwork(dt)
        net.dtype = dtype
        nodes = {}
        for lab, P in (("a", Pa), ("b", Pb)):
            nodes[lab] = rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype, train_params=list(names))
            net.add_node(lab, nodes[lab], node_type="diff_eq")
        net.add_func_node("out", kout, "identity")
        e_ab = net.add_edge("a", "b", weights=W_ab.copy(), train="gd", dtype=dtype)
        e_out = net.add_edge("b", "out", weights=W_out.copy(), train="gd", dtype=dtype)
        e_ba = net.add_edge("b", "a", weights=W_ba.copy(), feedback=True, train="gd", dtype=dtype)
        obs = net.fit_bptt(torch.tensor(x, dtype=dtype), torch.tensor(target, dtype=dtype), optimizer="sgd", lr=lr,
                           update_steps=win, sampling_steps=1, verbose=False)
        for lab in "ab":
            for key in names:
                out[f"rate_{lab}_{key.split('/')[-1]}_{tag}"] = np_(nodes[lab][key])
                out[f"rate_grad_{lab}_{key.split('/')[-1]}_{tag}"] = np_(nodes[lab][key].grad)  # (of the LAST window)
        out["rate_W_ab_" + tag], out["rate_W_ba_" + tag], out["rate_W_out_" + tag] = np_(e_ab.weights), np_(e_ba.weights), np_(e_out.weights)
        for key, e in (("W_ab", e_ab), ("W_ba", e_ba), ("W_out", e_out)):
            out[f"rate_grad_{key}_{tag}"] = np_(e.weights.grad)
        out["rate_loss_" + tag] = np.asarray([float(l) for l in obs["loss"]])
        out["rate_pred_" + tag] = obs.to_numpy("out")
    rate = dict(rate_x=x, rate_target=target, rate_W_ab0=W_ab, rate_W_ba0=W_ba, rate_W_out0=W_out,
                **{f"rate_a0_{k}": v for k, v in Pa.items()}, **{f"rate_b0_{k}": v for k, v in Pb.items()})
    # ---- (b) QIF pair -------------------------------------------------------------------------------------------
    nq, kq, Tq, winq, dtq, lrq = 36, 2, 100, 100, 1e-3, 1e-3
    r = np.random.default_rng(3710)
    Q = []
    for i in range(2):
        Q.append(dict(W=r.uniform(0, 1, (nq, nq)) * 20.0 / nq, eta=-5.0 + r.standard_normal(nq), tau=r.uniform(0.8, 1.2, nq),
                      k=r.uniform(0.8, 1.2, nq), tau_s=r.uniform(0.5, 1.5, nq),
                      y0=np.concatenate([r.uniform(-5.0, 70.0, nq), r.uniform(0.0, 1.0, nq)])))
    Wq_ab, Wq_ba = r.uniform(0, 1, (nq, nq)) * 10.0 / nq, -r.uniform(0, 1, (nq, nq)) * 10.0 / nq
    Wq_out = r.standard_normal((kq, nq)) / np.sqrt(nq)
    xq, targetq = 15.0 + r.standard_normal((Tq, nq)), r.standard_normal((Tq, kq)) * 0.1
    qnames = ["weights", "qif_op/tau", "qif_op/k", "qif_op/eta", "qif_op/tau_s"]
    for tag, dtype in dtypes.items():
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        net = FeedbackNetwork(dtq)
        net.dtype = dtype
        nodes = {}
        for lab, P in zip("ab", Q):
            args = (t(P["y0"]), torch.zeros(nq, dtype=dtype), torch.zeros(nq, dtype=dtype), t(P["W"]), t(P["tau"]), t(P["k"]),
                    t(P["eta"]), t(P["tau_s"]))
            pm = {"in": 0, "spike_var": 1, "weights": 2, "qif_op/tau": 3, "qif_op/k": 4, "qif_op/eta": 5, "qif_op/tau_s": 6}
            nodes[lab] = SpikeResetNet(make_qif(nq), args, {"out": (nq, 2 * nq), "reset_var": (0, nq)}, pm, spike_threshold=100.0,
                                       spike_reset=-100.0, dt=dtq, dtype=dtype, device="cpu", train_params=list(qnames))
            net.add_node(lab, nodes[lab], node_type="diff_eq")
        net.add_func_node("out", kq, "identity")
        e_ab = net.add_edge("a", "b", weights=Wq_ab.copy(), train="gd", dtype=dtype)
        e_out = net.add_edge("b", "out", weights=Wq_out.copy(), train="gd", dtype=dtype)
        e_ba = net.add_edge("b", "a", weights=Wq_ba.copy(), feedback=True, train="gd", dtype=dtype)
        spikes = []
        obs = net.fit_bptt(t(xq), t(targetq), optimizer="sgd", lr=lrq, update_steps=winq, sampling_steps=1, verbose=False)
        for lab in "ab":
            for key in qnames:
                out[f"qif_{lab}_{key.split('/')[-1]}_{tag}"] = np_(nodes[lab][key])
                out[f"qif_grad_{lab}_{key.split('/')[-1]}_{tag}"] = np_(nodes[lab][key].grad)
        out["qif_W_ab_" + tag], out["qif_W_ba_" + tag], out["qif_W_out_" + tag] = np_(e_ab.weights), np_(e_ba.weights), np_(e_out.weights)
        for key, e in (("W_ab", e_ab), ("W_ba", e_ba), ("W_out", e_out)):
            out[f"qif_grad_{key}_{tag}"] = np_(e.weights.grad)
        out["qif_loss_" + tag] = np.asarray([float(l) for l in obs["loss"]])
        out["qif_pred_" + tag] = obs.to_numpy("out")
    qif = dict(qif_x=xq, qif_target=targetq, qif_W_ab0=Wq_ab, qif_W_ba0=Wq_ba, qif_W_out0=Wq_out,
               **{f"qif_{lab}0_{k}": v for lab, P in zip("ab", Q) for k, v in P.items()})
    for kind in ("rate", "qif"):
        dev = max(np.abs(out[k] - out[k[:-3] + "f64"]).max() for k in out if k.startswith(kind) and k.endswith("f32") and "grad" not in k)
        gdev = max(np.abs(out[k] - out[k[:-3] + "f64"]).max() / np.abs(out[k[:-3] + "f64"]).max()
                   for k in out if k.startswith(kind + "_grad") and k.endswith("f32"))
        print(kind, "own float32 deviation of the reference: parameters / predictions", dev, "gradients (of each one's max)", gdev)
    print("QIF parameter movement: |W_ba - W_ba0| max", np.abs(out["qif_W_ba_f64"] - Wq_ba).max(),
          "|eta_a - eta_a0| max", np.abs(out["qif_a_eta_f64"] - Q[0]["eta"]).max())
    save("g37_feedback_fit", dt=dt, lr=lr, win=win, dtq=dtq, lrq=lrq, winq=winq, theta=100.0, v_reset=-100.0, slope=0.5,
         **rate, **qif, **out)


if __name__ == "__main__":
    main()
