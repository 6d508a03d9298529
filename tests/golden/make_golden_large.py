"""G23-G25: golden vectors at N >= 256 from the REFERENCE's own classes, so that the production code branches of the
kernels (full 256-column chunks with the staggered start, the column-strip adjoint, multi-tile MFMA GEMM, ragged last
chunk) are pinned to reference-derived numbers and not only to the co-designed oracle.

Run in the build container only:  python tests/golden/make_golden_large.py
(see make_golden.py for how the reference is imported and why the vector fields are hand-written).

Large arrays (W, the (T, N) inputs) are not stored: both sides regenerate them with tests/golden/recipes.py and the
fixture keeps their fingerprints.  All regenerated inputs are float32 values, so the reference's float64 run consumes
exactly the numbers the fp32 engine gets -- the difference that remains is arithmetic only.

  G23  BASELINE configs[0]: tanh rate network, N = 512, T = 2000, dt = 1e-2, YAML-default parameters, input
       0.5 * normal(T, N), Network.run(sampling_steps=1) (rectipy/network.py:542-601); g = 0.5 (full horizon) and
       g = 1.5 (chaotic: first 200 steps); plus N = 770 (n % 4 != 0: the 4-byte row-dot path), T = 100.
  G24  Network.fit_bptt, ONE SGD window, T = 100, N = 512 and N = 1028 (4 full chunks + a ragged 4-column chunk),
       all parameters trained; gradients read from the reference's `.grad` after the window
       (rectipy/network.py:1016-1048, 1123-1130).
  G25  SpikeResetNet + QIF, N = 512, T = 3000 (rectipy/nodes.py:382-392): spike (step, neuron) events.
"""
import numpy as np
import torch

import recipes
from make_golden import LI_PARAM_MAP, import_reference, li_tanh, make_qif, np_, rate_node, save


def fp(**arrays):
    return {"fp_" + k: recipes.fingerprint(v) for k, v in arrays.items()}


def main():
    Network, RateNet, SpikeResetNet, Spike = import_reference()
    dtypes = {"f64": torch.float64, "f32": torch.float32}

    # ---- G23 -------------------------------------------------------------------------------------------------
    dt = 1e-2
    for tag_g, n, g, T, keep, defaults in (("g05", 512, 0.5, 2000, 2000, True), ("g15", 512, 1.5, 2000, 200, True),
                                            ("n770", 770, 1.2, 100, 100, False)):
        P = recipes.li_network(n, seed=2300 + n, g=g, defaults=defaults)
        x = (0.5 * recipes.normal((T, n), seed=1)).astype(np.float32)
        cols = np.arange(0, n, 8)
        out = {}
        for tag, dtype in dtypes.items():
            net = Network(dt, dtype=dtype)
            net.add_node("rnn", rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype),
                         node_type="diff_eq")
            obs = net.run(x[:keep], sampling_steps=1, verbose=False, enable_grad=False)
            o = obs.to_numpy("out")
            assert o.shape == (keep, n) and list(obs["steps"]) == list(range(keep))
            out["out_" + tag] = o[:, cols].astype(np.float64 if tag == "f64" else np.float32)
            out["state_" + tag] = np_(net.state["rnn"])
        dev = np.abs(out["out_f64"] - out["out_f32"]).max()
        print(f"G23 {tag_g}: reference f32-vs-f64 max deviation {dev:.3e}, |out| max {np.abs(out['out_f64']).max():.3f}")
        save(f"g23_run_{tag_g}", dt=dt, n=n, g=g, T=T, keep=keep, seed=2300 + n, x_seed=1, defaults=defaults, cols=cols,
             ref_f32_dev=dev, **fp(W=P["W"], x=x), **out)

    # ---- G24 -------------------------------------------------------------------------------------------------
    for n in (512, 1028):
        T, kout, lr = 100, 3, 1e-2
        P = recipes.li_network(n, seed=2400 + n, g=1.5, defaults=False)
        W_out = (recipes.normal((kout, n), seed=2450 + n) / np.float32(np.sqrt(n))).astype(np.float32)
        x = (0.5 * recipes.normal((T, n), seed=2460 + n)).astype(np.float32)
        target = (0.1 * recipes.normal((T, kout), seed=2470 + n)).astype(np.float32)
        rows = np.array(sorted(set(range(0, n, 5)) | set(range(n - 12, n))))
        out = {}
        for tag, dtype in dtypes.items():
            net = Network(dt, dtype=dtype)
            node = rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype,
                             train_params=["weights", "li_op/tau", "li_op/k", "li_op/eta"])
            net.add_node("rnn", node, node_type="diff_eq")
            net.add_func_node("out", kout, "identity")
            edge = net.add_edge("rnn", "out", weights=W_out.copy(), train="gd")
            obs = net.fit_bptt(x, target, optimizer="sgd", lr=lr, update_steps=T, sampling_steps=1, verbose=False)
            out["loss_" + tag] = np.asarray([float(l) for l in obs["loss"]])
            out["pred_" + tag] = obs.to_numpy("out")
            out["state_" + tag] = np_(net.state["rnn"])
            if tag == "f64":
                gW = np_(node["weights"].grad)
                # the SGD update is what fit_bptt leaves behind: check that .grad is that window's gradient
                assert np.abs((P["W"].astype(np.float64) - np_(node["weights"])) / lr - gW).max() < 1e-10
                out["gW_rows"] = gW[rows].astype(np.float32)
                out["gW_rowsum"], out["gW_colsum"] = gW.sum(1), gW.sum(0)
                out["gW_absmax"] = np.abs(gW).max()
                if n == 512:
                    out["gW_full"] = gW.astype(np.float32)
                out["g_tau"], out["g_k"], out["g_eta"] = (np_(node[f"li_op/{q}"].grad) for q in ("tau", "k", "eta"))
                out["g_W_out"] = np_(edge.weights.grad)
                out["W_out_after"] = np_(edge.weights)
        save(f"g24_bptt_n{n}", dt=dt, lr=lr, n=n, T=T, seed=2400 + n, rows=rows,
             **fp(W=P["W"], x=x, W_out=W_out, target=target), **out)

    # ---- G25 -------------------------------------------------------------------------------------------------
    n, T, dt = 512, 3000, 1e-3
    P = recipes.qif_network(n, seed=2500)
    cols = np.arange(0, n, 8)
    out = {}
    for tag, dtype in dtypes.items():
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        args = (t(P["y0"]), torch.zeros(1, dtype=dtype), torch.zeros(n, dtype=dtype), t(P["W"]), t(P["tau"]), t(P["k"]),
                t(P["eta"]), t(P["tau_s"]))
        pm = {"in": 0, "spike_var": 1, "weights": 2, "qif_op/tau": 3, "qif_op/k": 4, "qif_op/eta": 5, "qif_op/tau_s": 6}
        vm = {"out": (n, 2 * n), "reset_var": (0, n), "qif_op/v": (0, n), "qif_op/s": (n, 2 * n)}
        node = SpikeResetNet(make_qif(n), args, vm, pm, spike_threshold=100.0, spike_reset=-100.0, dt=dt, dtype=dtype,
                             device="cpu")
        xin = torch.tensor([15.0], dtype=dtype)
        outs, events = [], []
        with torch.no_grad():
            for step in range(T):
                v_before = np_(node._y_spike).copy()
                o = np_(node.forward(xin))
                if step % 10 == 0:
                    outs.append(o[cols].copy())
                for idx in np.nonzero(v_before >= 100.0)[0]:
                    events.append((step, idx))
        out["s_" + tag] = np.stack(outs)
        out["events_" + tag] = np.asarray(events, dtype=np.int32).reshape(-1, 2)
        out["v_final_" + tag] = np_(node._y_spike)
        out["s_final_" + tag] = np_(node._y_stop)
        print(f"G25 {tag}: {len(events)} spikes")
    same = out["events_f64"].shape == out["events_f32"].shape and (out["events_f64"] == out["events_f32"]).all()
    print("G25: f32 and f64 reference spike events identical:", same)
    save("g25_qif_n512", dt=dt, n=n, T=T, seed=2500, x_const=15.0, theta=100.0, v_reset=-100.0, stride=10, cols=cols,
         events_agree=same, **fp(W=P["W"], eta=P["eta"]), **out)


if __name__ == "__main__":
    main()
