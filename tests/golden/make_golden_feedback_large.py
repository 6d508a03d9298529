"""G36: FeedbackNetwork.run by the REFERENCE (rectipy/network.py:1196-1357) at sizes where the coupled step kernel's
16-byte paths are the code that runs (rp_forward_coupled: full 256-column chunks + a ragged one in both the recurrent and
the edge matrices, different widths for the two populations):
  (a) rate a (N = 260) -> rate b (N = 516), feedback b -> a, input of width 260, T = 60;
  (b) QIF a -> QIF b (N = 260 each) with inhibitory feedback b -> a, constant input 15, T = 1500 (spikes in both).
W, the edges and the inputs are regenerated from tests/golden/recipes.py on both sides (their fingerprints are stored);
the fixture holds the reference's outputs in float64 and float32.
Run in the build container only:  python tests/golden/make_golden_feedback_large.py"""
import numpy as np
import torch

import recipes
from make_golden import import_reference, make_qif, np_, rate_node, save


def edge(rows, cols, seed, g):
    return (recipes.normal((rows, cols), seed) * np.float32(g / np.sqrt(cols))).astype(np.float32)


def main():
    Network, RateNet, SpikeResetNet, Spike = import_reference()
    from rectipy.network import FeedbackNetwork
    dtypes = {"f64": torch.float64, "f32": torch.float32}
    out, fps = {}, {}
    # ---- (a) rate populations of different widths ---------------------------------------------------------------
    na, nb, T, dt = 260, 516, 60, 1e-2
    Pa = recipes.li_network(na, seed=3601, g=0.9, defaults=False)
    Pb = recipes.li_network(nb, seed=3602, g=0.9, defaults=False)
    W_ab, W_ba = edge(nb, na, 3603, 0.6), edge(na, nb, 3604, 0.6)
    x = (0.5 * recipes.normal((T, na), seed=3605)).astype(np.float32)
    for name, arr in (("Wa", Pa["W"]), ("Wb", Pb["W"]), ("W_ab", W_ab), ("W_ba", W_ba), ("x", x)):
        fps["fp_" + name] = recipes.fingerprint(arr)
    for tag, dtype in dtypes.items():
        net = FeedbackNetwork(dt)
        net.dtype = dtype
        net.add_node("a", rate_node(RateNet, Pa["W"], Pa["tau"], Pa["k"], Pa["eta"], Pa["y0"], dt, dtype), node_type="diff_eq")
        net.add_node("b", rate_node(RateNet, Pb["W"], Pb["tau"], Pb["k"], Pb["eta"], Pb["y0"], dt, dtype), node_type="diff_eq")
        net.add_edge("a", "b", weights=W_ab.astype(np.float64), dtype=dtype)
        net.add_edge("b", "a", weights=W_ba.astype(np.float64), feedback=True, dtype=dtype)
        obs = net.run(torch.tensor(x, dtype=dtype), sampling_steps=1, verbose=False, enable_grad=False)
        out["rate_out_" + tag] = obs.to_numpy("out")
        out["rate_state_a_" + tag], out["rate_state_b_" + tag] = np_(net.state["a"]), np_(net.state["b"])
    # ---- (b) QIF pair -------------------------------------------------------------------------------------------
    nq, Tq, dtq = 260, 1500, 1e-3
    Wq = [(recipes.uniform((nq, nq), 3610 + i) * np.float32(20.0 / nq)).astype(np.float32) for i in range(2)]
    etaq = [(-5.0 + recipes.normal((nq,), 3612)).astype(np.float32), (12.0 + recipes.normal((nq,), 3613)).astype(np.float32)]
    Wq_ab = (recipes.uniform((nq, nq), 3614) * np.float32(10.0 / nq)).astype(np.float32)
    Wq_ba = (-recipes.uniform((nq, nq), 3615) * np.float32(10.0 / nq)).astype(np.float32)
    y0 = np.concatenate([np.full(nq, -2.0), np.zeros(nq)])
    for name, arr in (("Wq0", Wq[0]), ("Wq1", Wq[1]), ("etaq0", etaq[0]), ("etaq1", etaq[1]), ("Wq_ab", Wq_ab), ("Wq_ba", Wq_ba)):
        fps["fp_" + name] = recipes.fingerprint(arr)
    for tag, dtype in dtypes.items():
        t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), dtype=dtype)

        def qif(i):
            one = np.ones(nq)
            args = (t(y0), torch.zeros(nq, dtype=dtype), torch.zeros(nq, dtype=dtype), t(Wq[i]), t(one), t(one), t(etaq[i]), t(one))
            return SpikeResetNet(make_qif(nq), args, {"out": (nq, 2 * nq), "reset_var": (0, nq)},
                                 {"in": 0, "spike_var": 1, "weights": 2}, spike_threshold=100.0, spike_reset=-100.0, dt=dtq,
                                 dtype=dtype, device="cpu")
        net = FeedbackNetwork(dtq)
        net.dtype = dtype
        net.add_node("a", qif(0), node_type="diff_eq")
        net.add_node("b", qif(1), node_type="diff_eq")
        net.add_edge("a", "b", weights=Wq_ab.astype(np.float64), dtype=dtype)
        net.add_edge("b", "a", weights=Wq_ba.astype(np.float64), feedback=True, dtype=dtype)
        o = net.run(torch.full((Tq, nq), 15.0, dtype=dtype), sampling_steps=1, verbose=False, enable_grad=False).to_numpy("out")
        out["qif_out_" + tag] = o[::10]
        out["qif_peak_" + tag] = o.max()
        out["qif_final_" + tag] = o[-1]
    print("own float32 deviation of the reference: rate", np.abs(out["rate_out_f32"] - out["rate_out_f64"]).max(),
          "qif", np.abs(out["qif_out_f32"] - out["qif_out_f64"]).max())
    save("g36_feedback_large", dt=dt, dtq=dtq, T=T, Tq=Tq, na=na, nb=nb, nq=nq, **fps, **out)


if __name__ == "__main__":
    main()
