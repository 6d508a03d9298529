"""G27 / G28: edge variants and FeedbackNetwork captured from the REFERENCE's own classes
(rectipy/edges.py:68-174: LinearMasked / LinearMemory / LinearFilter / LinearMemoryFilter; rectipy/network.py:1196-1357).
Run in the build container only:  python tests/golden/make_golden_topologies.py
(see make_golden.py for how the reference is imported and why the vector fields are hand-written).

G27a  the four edge classes driven directly with an input sequence (non-square and square weights / masks, repeated
      delays) -- pins orientation rules (square matrices ARE transposed, edges.py:22-26, 161-162), the delay-buffer
      indexing (edges.py:91-94) and the filter recursions;
G27b  the same edges inside Network.run: inp -(masked)-> rnn -(filter)-> out, and rnn -(memory / memory+filter)-> out;
G28   FeedbackNetwork.run: (a) rate a -> rate b with feedback b -> a (square edges), (b) QIF a -> QIF b with feedback
      b -> a, (c) chain a -> b -> c with a feedback edge a -> c (source already stepped).  The reference cannot sum
      several inputs of one node (network.py:1333,1344 build a tensor from a list of tensors), so every node has one
      feed-forward and at most one feedback input; feedback from an activation node fails in the reference
      (InstantNode["out"] is None) and is not captured.
"""
import numpy as np
import torch

from make_golden import import_reference, make_qif, np_, rate_node, rng_li, save


def main():
    Network, RateNet, SpikeResetNet, Spike = import_reference()
    from rectipy.edges import LinearFilter, LinearMasked, LinearMemory, LinearMemoryFilter
    from rectipy.network import FeedbackNetwork
    dtypes = {"f64": torch.float64, "f32": torch.float32}

    # ---- G27a ------------------------------------------------------------------------------------------------
    r = np.random.default_rng(270)
    T, n_in, n_out = 14, 6, 4
    x = r.standard_normal((T, n_in))
    xs = r.standard_normal((T, 5))
    W, Ws = r.standard_normal((n_out, n_in)), r.standard_normal((5, 5))
    Wt = np.ascontiguousarray(W.T)  # (n_in, n_out) orientation: transposed by the constructor
    mask = (r.uniform(size=(n_out, n_in)) > 0.4).astype(np.float64)
    mask_s = (r.uniform(size=(5, 5)) > 0.4).astype(np.float64)
    delays = np.array([2, 0, 3, 1, 3, 1])
    F = r.standard_normal((n_in, n_in)) * 0.3
    out = {}
    for tag, dtype in dtypes.items():
        def seq(edge, data):
            with torch.no_grad():
                return np.stack([np_(edge.forward(torch.tensor(row, dtype=dtype))) for row in data])
        out["masked_" + tag] = seq(LinearMasked(n_in, n_out, mask=mask, weights=W, dtype=dtype), x)
        out["masked_T_" + tag] = seq(LinearMasked(n_in, n_out, mask=np.ascontiguousarray(mask.T), weights=Wt, dtype=dtype), x)
        out["masked_sq_" + tag] = seq(LinearMasked(5, 5, mask=mask_s, weights=Ws, dtype=dtype), xs)
        out["memory_" + tag] = seq(LinearMemory(n_in, n_out, delays=delays, weights=W, dtype=dtype), x)
        out["filter_" + tag] = seq(LinearFilter(n_in, n_out, filter_weights=F, weights=W, dtype=dtype), x)
        out["memfilter_" + tag] = seq(LinearMemoryFilter(n_in, n_out, delays=delays, filter_weights=F, weights=W,
                                                         dtype=dtype), x)
    save("g27a_edge_variants", x=x, xs=xs, W=W, Wt=Wt, Ws=Ws, mask=mask, mask_s=mask_s, delays=delays, F=F, **out)

    # ---- G27b ------------------------------------------------------------------------------------------------
    n, m, k, T, dt = 12, 4, 3, 30, 1e-2
    P = rng_li(n, seed=271)
    r = np.random.default_rng(272)
    W_in, W_out = r.standard_normal((n, m)), r.standard_normal((k, n))
    mask = (r.uniform(size=(n, m)) > 0.5).astype(np.float64)
    F = r.standard_normal((n, n)) * 0.1
    delays = np.array([0, 1, 2, 0, 1, 2, 3, 3, 0, 1, 2, 3])
    x_m, x_n = r.standard_normal((T, m)), r.standard_normal((T, n))
    out = {}
    for tag, dtype in dtypes.items():
        def rnn():
            return rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype)
        net = Network(dt, dtype=dtype)
        net.add_func_node("inp", m, "identity")
        net.add_node("rnn", rnn(), node_type="diff_eq")
        net.add_func_node("out", k, "identity")
        net.add_edge("inp", "rnn", weights=W_in, mask=mask)
        net.add_edge("rnn", "out", weights=W_out, filter_weights=F)
        out["masked_filter_" + tag] = net.run(x_m, sampling_steps=1, verbose=False, enable_grad=False).to_numpy("out")
        for name, kw in (("memory", dict(delays=delays)), ("memfilter", dict(delays=delays, filter_weights=F))):
            net = Network(dt, dtype=dtype)
            net.add_node("rnn", rnn(), node_type="diff_eq")
            net.add_func_node("out", k, "identity")
            net.add_edge("rnn", "out", weights=W_out, **kw)
            out[f"{name}_{tag}"] = net.run(torch.tensor(x_n, dtype=dtype), sampling_steps=1, verbose=False,
                                           enable_grad=False).to_numpy("out")
    save("g27b_edge_variants_in_network", dt=dt, x_m=x_m, x_n=x_n, W_in=W_in, W_out=W_out, mask=mask, F=F, delays=delays,
         **P, **out)

    # ---- G28 -------------------------------------------------------------------------------------------------
    n, T, dt = 10, 40, 1e-2
    Pa, Pb, Pc = rng_li(n, seed=281), rng_li(n, seed=282), rng_li(n, seed=283)
    r = np.random.default_rng(284)
    W_ab, W_bc = r.standard_normal((n, n)) * 0.5, r.standard_normal((n, n)) * 0.5
    W_ba, W_ac = r.standard_normal((n, n)) * 0.5, r.standard_normal((n, n)) * 0.5
    x = r.standard_normal((T, n))
    out = {}
    for tag, dtype in dtypes.items():
        def rate(P):
            return rate_node(RateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, dtype)
        # (a) a -> b, feedback b -> a
        net = FeedbackNetwork(dt)
        net.dtype = dtype
        net.add_node("a", rate(Pa), node_type="diff_eq")
        net.add_node("b", rate(Pb), node_type="diff_eq")
        net.add_edge("a", "b", weights=W_ab, dtype=dtype)
        net.add_edge("b", "a", weights=W_ba, feedback=True, dtype=dtype)
        obs = net.run(torch.tensor(x, dtype=dtype), sampling_steps=1, verbose=False, enable_grad=False)
        out["rate_ab_" + tag] = obs.to_numpy("out")
        out["rate_ab_state_a_" + tag], out["rate_ab_state_b_" + tag] = np_(net.state["a"]), np_(net.state["b"])
        # (c) a -> b -> c, feedback edge a -> c (its source is stepped before c is evaluated)
        net = FeedbackNetwork(dt)
        net.dtype = dtype
        for lab, P in (("a", Pa), ("b", Pb), ("c", Pc)):
            net.add_node(lab, rate(P), node_type="diff_eq")
        net.add_edge("a", "b", weights=W_ab, dtype=dtype)
        net.add_edge("b", "c", weights=W_bc, dtype=dtype)
        net.add_edge("a", "c", weights=W_ac, feedback=True, dtype=dtype)
        out["rate_abc_" + tag] = net.run(torch.tensor(x, dtype=dtype), sampling_steps=1, verbose=False,
                                         enable_grad=False).to_numpy("out")
    # (b) spiking pair
    nq, Tq, dtq = 16, 2500, 1e-3
    rq = np.random.default_rng(285)
    Wq = [rq.uniform(0, 1, (nq, nq)) * 20.0 / nq for _ in range(2)]
    etaq = [-5.0 + rq.standard_normal(nq), 12.0 + rq.standard_normal(nq)]  # b gets no external drive: excitable by itself
    Wq_ab, Wq_ba = rq.uniform(0, 1, (nq, nq)) * 10.0 / nq, -rq.uniform(0, 1, (nq, nq)) * 10.0 / nq
    y0 = np.concatenate([np.full(nq, -2.0), np.zeros(nq)])
    for tag, dtype in dtypes.items():
        t = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        def qif(i):
            one = np.ones(nq)
            args = (t(y0), torch.zeros(nq, dtype=dtype), torch.zeros(nq, dtype=dtype), t(Wq[i]), t(one), t(one), t(etaq[i]),
                    t(one))
            pm = {"in": 0, "spike_var": 1, "weights": 2}
            vm = {"out": (nq, 2 * nq), "reset_var": (0, nq)}
            return SpikeResetNet(make_qif(nq), args, vm, pm, spike_threshold=100.0, spike_reset=-100.0, dt=dtq,
                                 dtype=dtype, device="cpu")
        net = FeedbackNetwork(dtq)
        net.dtype = dtype
        net.add_node("a", qif(0), node_type="diff_eq")
        net.add_node("b", qif(1), node_type="diff_eq")
        net.add_edge("a", "b", weights=Wq_ab, dtype=dtype)
        net.add_edge("b", "a", weights=Wq_ba, feedback=True, dtype=dtype)
        xq = torch.full((Tq, nq), 15.0, dtype=dtype)
        o = net.run(xq, sampling_steps=1, verbose=False, enable_grad=False).to_numpy("out")
        out["qif_ab_" + tag] = o[::5]
        out["qif_ab_peak_" + tag] = o.max()
    save("g28_feedback", dt=dt, x=x, W_ab=W_ab, W_bc=W_bc, W_ba=W_ba, W_ac=W_ac,
         **{f"{k_}_{lab}": v for lab, P in (("a", Pa), ("b", Pb), ("c", Pc)) for k_, v in P.items()},
         dtq=dtq, Tq=Tq, Wq0=Wq[0], Wq1=Wq[1], etaq0=etaq[0], etaq1=etaq[1], Wq_ab=Wq_ab, Wq_ba=Wq_ba, y0q=y0, **out)


if __name__ == "__main__":
    main()
