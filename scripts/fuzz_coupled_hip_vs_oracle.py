"""GPU box: FeedbackNetwork graphs and Network.run_batch on the HIP library against the same package on the CPU oracle, on
RANDOM shapes -- 2..4 populations (tanh / sigmoid / QIF / LIF / QIF_SFA / IK, sizes from 5 to ~1100: 4-byte and 16-byte paths, ragged
chunks, different widths), random feed-forward chains with random feedback edges (the three edge read modes of
rp_forward_coupled arise from their direction and the source's kind), optional input / read-out layers, recorded
variables, sampling, cutoff, two consecutive runs, in a third of the cases with every matrix thinned to a random density and
streamed packed (populations and edges: rp_run.edge_sp); the same graphs with every parameter trained (fit_bptt, two SGD windows:
the fused adjoint rp_backward_coupled on the GPU and on the oracle against autograd over the per-step walk, to 1e-4, 5e-3
with spiking populations); run_batch with random B, sampling and input edge.  Populations'
outputs are compared BIT FOR BIT where no device-side torch product lies in front of them (no input layer), else to 2e-5
(5e-4 for spiking populations, whose spike times may move by a step behind a rounded input); batched runs bit for bit
up to 4 trajectories (vector ALUs) and to 2e-5 on the matrix cores.

    python scripts/fuzz_coupled_hip_vs_oracle.py [cases] [seed] [max_n]
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.backend import OracleBackend  # noqa: E402
from rectipy_amd import FeedbackNetwork, Network  # noqa: E402

DEV = os.environ.get("FUZZ_DEVICE", "cuda:0")
P = "neuron_model_templates."
KINDS = {
    "tanh": dict(path=P + "rate_neurons.leaky_integrator.tanh", op="li_op", src="tanh_op/r", tgt="li_op/r_in", outs=["li_op/v", "tanh_op/r"]),
    "sigmoid": dict(path=P + "rate_neurons.leaky_integrator.sigmoid", op="li_op", src="sigmoid_op/r", tgt="li_op/r_in", outs=["li_op/v"]),
    "qif": dict(path=P + "spiking_neurons.qif.qif", op="qif_op", src="qif_op/s", tgt="qif_op/s_in", outs=["qif_op/s"], spike=True),
    "lif": dict(path=P + "spiking_neurons.lif.lif", op="lif_op", src="lif_op/s", tgt="lif_op/s_in", outs=["lif_op/s"], spike=True),
    "qif_sfa": dict(path=P + "spiking_neurons.qif.qif_sfa", op="qif_sfa_op", src="qif_sfa_op/s", tgt="qif_sfa_op/s_in",
                    outs=["qif_sfa_op/s"], spike=True),
    "ik": dict(path=P + "spiking_neurons.ik.ik", op="ik_op", src="ik_op/s", tgt="ik_op/s_in", outs=["ik_op/s"], spike=True),
}
TRAIN = {"tanh": ["tau", "k", "eta"], "sigmoid": ["tau", "k", "eta"], "qif": ["tau", "k", "eta", "tau_s"], "lif": ["tau", "k", "eta", "tau_s"],
         "qif_sfa": ["tau", "k", "eta", "tau_s", "alpha", "tau_x"],
         "ik": ["C", "k", "eta", "tau_s", "g", "b", "tau_u", "kappa", "v_r", "v_theta", "E_r"]}


def add_population(net, label, kind, n, rng, train=False, density=None, train_weights=True):
    K = KINDS[kind]
    op = K["op"]
    if K.get("spike"):
        W = np.abs(rng.standard_normal((n, n))) * 2.0 / n
        lo_hi = {"qif": dict(tau=(0.8, 1.2), k=(0.5, 1.5), eta=(0.5, 3.0), tau_s=(0.5, 2.0)),
                 "lif": dict(tau=(0.4, 0.8), k=(0.5, 1.5), eta=(2.0, 5.0), tau_s=(0.2, 0.6)),
                 "qif_sfa": dict(tau=(0.8, 1.2), k=(0.5, 1.5), eta=(0.5, 3.0), tau_s=(0.5, 2.0), alpha=(0.2, 1.0), tau_x=(2.0, 8.0)),
                 "ik": dict(C=(80.0, 120.0), k=(0.5, 0.9), eta=(40.0, 90.0), tau_s=(4.0, 8.0), g=(0.5, 1.5), b=(-3.0, -1.0),
                            tau_u=(20.0, 40.0), kappa=(5.0, 15.0), v=(-65.0, -35.0), u=(0.0, 5.0))}[kind]
        node_vars = {f"all/{op}/{q}": rng.uniform(lo, hi, n) for q, (lo, hi) in lo_hi.items()}
        thr = {"qif": (1.0, -1.0), "lif": (1.0, 0.0), "qif_sfa": (1.0, -1.0), "ik": (-36.0, -60.0)}[kind]  # (ik: some start above it)
        kw = dict(spike_var=f"{op}/spike", reset_var=f"{op}/v", spike_threshold=thr[0], spike_reset=thr[1])
    else:
        W = rng.standard_normal((n, n)) * 1.0 / np.sqrt(n)
        node_vars = {"all/li_op/tau": rng.uniform(0.5, 2.0, n), "all/li_op/k": rng.uniform(0.5, 1.5, n),
                     "all/li_op/eta": rng.standard_normal(n) * 0.1, "all/li_op/v": rng.standard_normal(n) * 0.5}
        kw = {}
    if density is not None:  # a fixed sparse matrix: streamed packed where the 16-byte path exists (forced at any size)
        W = W * (rng.uniform(size=W.shape) < density) / density ** 0.5
        if n % 4 == 0:
            kw["sparse_weights"] = True
    out_var = K["outs"][int(rng.integers(0, len(K["outs"])))]
    if train:
        kw["train_params"] = (["weights"] if train_weights else []) + [f"{op}/{q}" for q in TRAIN[kind]]
    net.add_diffeq_node(label, K["path"], weights=W, source_var=K["src"], target_var=K["tgt"], input_var=f"{op}/I_ext",
                        output_var=out_var, dtype=torch.float32, node_vars=node_vars, **kw)
    return f"{op}/v"


def build_feedback(where, seed, kinds, sizes, m, k, fb_pairs, train=False, density=None, fixed_edges=False):
    rng = np.random.default_rng(seed)
    net = FeedbackNetwork(5e-3, **where)
    net.PACKED_EDGE_MIN_ROWS = 0   # (sparse edges are packed at any size here)
    if m:
        net.add_func_node("inp", m, "identity")
    labels = "abcd"[:len(sizes)]
    rec_var = None
    for lab, kind, n in zip(labels, kinds, sizes):
        # fixed_edges == 2: the populations' own matrices are fixed and sparse too (only per-neuron parameters trained)
        v = add_population(net, lab, kind, n, rng, train, density if (not fixed_edges or fixed_edges == 2) else None,
                           train_weights=fixed_edges != 2)
        rec_var = rec_var or (lab, v)
    if k:
        net.add_func_node("out", k, ["identity", "tanh"][int(rng.integers(0, 2))])
    order = (["inp"] if m else []) + list(labels) + (["out"] if k else [])
    width = dict(zip(labels, sizes), inp=m, out=k)
    thin = lambda W: W if density is None else W * (rng.uniform(size=W.shape) < density) / density ** 0.5
    for s_, t_ in zip(order[:-1], order[1:]):
        between = s_ in labels and t_ in labels
        net.add_edge(s_, t_, weights=thin(rng.standard_normal((width[t_], width[s_])) * 0.7 / np.sqrt(width[s_])),
                     train="gd" if train and not (fixed_edges and between) else None)
    for s_, t_ in fb_pairs:
        net.add_edge(labels[s_], labels[t_], weights=thin(rng.standard_normal((sizes[t_], sizes[s_])) * 0.5 / np.sqrt(sizes[s_])), feedback=True,
                     train="gd" if train and not fixed_edges else None)
    return net, rec_var


def vary_edges(net, labels, sizes, fb_pairs, rng, train=False, kinds=("plain", "delay", "filter", "delay_filter", "masked", "node"), many=True):
    """r05: replace edges between populations by the variants that used to force the per-step walk -- edges with memory
    (delay / filter / delay + filter), masked edges, an element-wise activation node in between -- and, rarely, pile more than
    RP_MAX_EDGES edges onto one population through small side populations.  Returns what was done (for the failure report)."""
    done = []
    pairs = [(labels[i], labels[i + 1], False) for i in range(len(labels) - 1)] + [(labels[s_], labels[t_], True) for s_, t_ in fb_pairs]
    width = dict(zip(labels, sizes))
    for s_, t_, fb in pairs:
        kind = kinds[int(rng.integers(0, len(kinds)))]
        if kind == "plain" or (fb and kind == "node"):
            continue
        ns, nt = width[s_], width[t_]
        W = rng.standard_normal((nt, ns)) * 0.6 / np.sqrt(ns)
        net.pop_edge(s_, t_)
        kw = dict(feedback=True) if fb else {}
        if train:
            kw["train"] = "gd"
        if kind == "delay":
            net.add_edge(s_, t_, weights=W, delays=rng.integers(0, 5, size=ns), **kw)
        elif kind == "filter":
            net.add_edge(s_, t_, weights=W, filter_weights=rng.standard_normal((ns, ns)) * 0.5 / np.sqrt(ns), **kw)
        elif kind == "delay_filter":
            net.add_edge(s_, t_, weights=W, delays=rng.choice([1, 3, 4], size=ns), filter_weights=rng.standard_normal((ns, ns)) * 0.5 / np.sqrt(ns), **kw)
        elif kind == "masked":
            net.add_edge(s_, t_, weights=W, mask=(rng.uniform(size=(nt, ns)) < 0.5).astype(np.float32), **kw)
        else:
            w = int(rng.integers(3, 70))
            name = f"f_{s_}{t_}"
            net.add_func_node(name, w, ["tanh", "sigmoid", "identity"][int(rng.integers(0, 3))])
            net.add_edge(s_, name, weights=rng.standard_normal((w, ns)) * 0.7 / np.sqrt(ns), train="gd" if train else None)
            net.add_edge(name, t_, weights=rng.standard_normal((nt, w)) * 0.7 / np.sqrt(w), train="gd" if train else None)
        done.append(f"{s_}->{t_}:{kind}")
    if many and rng.integers(0, 5) == 0 and len(labels) >= 2:   # ten more edges into the last population
        t_ = labels[-1]
        for q in range(10):
            add_population(net, f"p{q}", "tanh", 6, rng)
            net.add_edge(labels[0], f"p{q}", weights=rng.standard_normal((6, sizes[0])) * 0.4 / np.sqrt(sizes[0]))
            net.add_edge(f"p{q}", t_, weights=rng.standard_normal((sizes[-1], 6)) * 0.4)
        done.append(f"{t_}:11 edges")
    return done


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and bool(((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))).all())


def close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())) if a.size else a.shape == b.shape


def small4(sizes):
    """FUZZ_SMALL4=1: every population on the 16-byte path with at most 512 neurons -- the shapes whose rp_forward_coupled calls run
    all their steps in ONE resident launch (r05)."""
    if os.environ.get("FUZZ_SMALL4"):
        return [min(1024 if len(sizes) == 2 else 512, max(8, n - n % 4)) for n in sizes]
    return sizes


def main(cases=40, seed=0, max_n=1100):
    rng = np.random.default_rng(seed)
    fails = 0
    wheres = (dict(device="cpu", backend=OracleBackend()), dict(device=DEV) if DEV != "cpu" else dict(device="cpu", backend=OracleBackend()))
    for case in range(cases):
        cfg = {}
        try:
            mode = int(rng.integers(0, 6))
            if mode >= 4:  # ---- r05: graphs with edges with memory / masks / activation nodes between populations / long edge lists ----
                npop = int(rng.integers(2, 4))
                kinds = [["tanh", "sigmoid", "qif", "tanh"][int(rng.integers(0, 4))] for _ in range(npop)]
                sizes = [int(rng.choice([rng.integers(5, 40), rng.integers(60, 70), rng.integers(250, 300), rng.integers(min(500, max_n - 1), max_n)]))
                         for _ in range(npop)]
                sizes = small4(sizes)
                m, k = int(rng.integers(0, 3)) * int(rng.integers(0, 2)), int(rng.integers(0, 3))
                pairs = {(npop - 1, 0)} if rng.integers(0, 4) else set()
                T = int(rng.integers(12, 40))
                samp, cutoff = int(rng.integers(1, 4)), int(rng.integers(0, 4))
                cfg = dict(kind="variants", kinds=kinds, sizes=sizes, m=m, k=k, fb=sorted(pairs), T=T, samp=samp, cutoff=cutoff)
                x = (rng.standard_normal((T, m if m else sizes[0])) * 0.5 + (1.5 if KINDS[kinds[0]].get("spike") else 0.0)).astype(np.float32)
                res = []
                for where, walk in ((wheres[0], True), (wheres[0], False), (wheres[1], False)):
                    with contextlib.redirect_stdout(io.StringIO()):
                        net, _ = build_feedback(where, 1000 * seed + case, kinds, sizes, m, k, sorted(pairs))
                        cfg["edges"] = vary_edges(net, "abcd"[:npop], sizes, sorted(pairs), np.random.default_rng(7000 * seed + case))
                        net._force_stepwise = walk
                        net.compile()
                        planned = net._coupled_plan([]) is not None
                        o1 = net.run(x[:T // 2], sampling_steps=samp, cutoff=cutoff, verbose=False, enable_grad=False)
                        o2 = net.run(x[T // 2:], sampling_steps=1, verbose=False, enable_grad=False)
                    mem = {}
                    for s_, t_ in net.graph.edges:
                        for name in ("buffer", "y"):
                            if hasattr(net.get_edge(s_, t_), name):
                                mem[f"{s_}{t_}{name}"] = getattr(net.get_edge(s_, t_), name).detach().cpu().numpy()
                    res.append((o1.to_numpy("out"), o2.to_numpy("out"), {q: v.detach().cpu().numpy() for q, v in net.state.items()}, mem, planned))
                spiking = any(KINDS[q].get("spike") for q in kinds)
                tol = 5e-4 if spiking else 2e-5
                # bits only where no device-side torch product lies in front of a population: no input / read-out layer, and a
                # graph WITH a feedback edge (without one the window-by-window plan evaluates the edges as torch products)
                exact = m == 0 and k == 0 and bool(pairs)
                ok = (not res[0][4]) and res[1][4] == res[2][4] and (res[1][4] or not any(cfg["edges"]))
                for other in (1, 2):
                    same = bits_equal if (other == 2 and exact) else (lambda u, v: close(u, v, tol))
                    base = res[1] if other == 2 else res[0]     # HIP against the oracle's plan; the oracle's plan against its walk
                    ok = ok and same(base[0], res[other][0]) and same(base[1], res[other][1])
                    for q in base[2]:
                        ok = ok and same(base[2][q], res[other][2][q])
                    for q in base[3]:
                        ok = ok and same(base[3][q], res[other][3][q])
                cfg["coupled_plan"] = res[1][4]
            elif mode == 3:  # ---- a FeedbackNetwork trained through its cycles: fit_bptt, every parameter ----
                npop = int(rng.integers(2, 4))
                kinds = [list(KINDS)[int(rng.integers(0, len(KINDS)))] for _ in range(npop)]
                sizes = [int(rng.choice([rng.integers(5, 40), rng.integers(250, 300), rng.integers(min(500, max_n - 1), max_n)])) for _ in range(npop)]
                sizes = small4(sizes)
                m, k = int(rng.integers(0, 3)), int(rng.integers(1, 3))
                pairs = set()
                for _ in range(int(rng.integers(1, 3))):
                    s_, t_ = int(rng.integers(0, npop)), int(rng.integers(0, npop))
                    if s_ != t_ and (s_, t_) not in pairs and t_ != s_ + 1:
                        pairs.add((s_, t_))
                if not pairs:
                    pairs = {(npop - 1, 0)}
                win = int(rng.integers(8, 25))
                T = 2 * win + int(rng.integers(0, 5))
                spiking = any(KINDS[q].get("spike") for q in kinds)
                # a third of the cases: trained populations, FIXED sparse edges between them (packed in forward and adjoint on the GPU)
                fixed = int(rng.integers(0, 3) == 0) * int(rng.integers(1, 3))   # 0 | 1: fixed sparse edges | 2: ... and fixed sparse W
                density = float(rng.uniform(0.03, 0.12)) if fixed else None
                if fixed:
                    sizes = [max(8, n - n % 4) for n in sizes]
                varied = (not fixed) and int(rng.integers(0, 2)) == 0
                cfg = dict(kind="fit", kinds=kinds, sizes=sizes, m=m, k=k, fb=sorted(pairs), T=T, win=win, fixed_sparse=int(fixed), varied=varied)
                x = (rng.standard_normal((T, m if m else sizes[0])) * 0.5 + (1.5 if KINDS[kinds[0]].get("spike") else 0.0)).astype(np.float32)
                y = (rng.standard_normal((T, k)) * 0.2).astype(np.float32)
                res = []
                for where, walk in ((wheres[0], True), (wheres[0], False), (wheres[1], False)):
                    with contextlib.redirect_stdout(io.StringIO()):
                        net, _ = build_feedback(where, 1000 * seed + case, kinds, sizes, m, k, sorted(pairs), train=True, density=density,
                                                fixed_edges=int(fixed))
                        if varied:   # r05 (ABI v18): trained edges with memory / masks / activation nodes inside the cycle
                            cfg["edges"] = vary_edges(net, "abcd"[:npop], sizes, sorted(pairs), np.random.default_rng(9000 * seed + case), train=True,
                                                      kinds=("plain", "delay", "filter", "masked", "node"), many=False)
                        net._force_stepwise_adjoint = walk
                        net.compile()
                        fused = net._coupled_grad_plan() is not None
                        obs = net.fit_bptt(x, y, optimizer="sgd", lr=2e-3 if spiking else 2e-2, update_steps=win, sampling_steps=1, verbose=False)
                    uniq = list({id(q): q for q in net.parameters()}.values())
                    res.append(([q.detach().cpu().numpy() for q in uniq], obs.to_numpy("out"), fused))
                tol = 5e-3 if spiking else 1e-4
                ok = res[1][2] == res[2][2] and not res[0][2]
                for other in (1, 2):   # oracle fused vs oracle walk; HIP fused vs oracle walk
                    for a, b in zip(res[0][0], res[other][0]):
                        ok = ok and close(a, b, tol)
                    ok = ok and close(res[0][1], res[other][1], tol)
                cfg["fused_adjoint"] = res[1][2]
            elif mode < 2:  # ---- a FeedbackNetwork ----
                npop = int(rng.integers(2, 5))
                kinds = [list(KINDS)[int(rng.integers(0, len(KINDS)))] for _ in range(npop)]
                sizes = [int(rng.choice([rng.integers(5, 40), rng.integers(250, 300), rng.integers(min(500, max_n - 1), max_n)])) for _ in range(npop)]
                sizes = small4(sizes)
                m, k = int(rng.integers(0, 3)) * int(rng.integers(0, 2)), int(rng.integers(0, 3))
                pairs = set()
                for _ in range(int(rng.integers(1, 4))):
                    s_, t_ = int(rng.integers(0, npop)), int(rng.integers(0, npop))
                    if s_ != t_ and (s_, t_) not in pairs and t_ != s_ + 1:
                        pairs.add((s_, t_))
                if not pairs:
                    pairs = {(npop - 1, 0)}
                T = int(rng.integers(20, 70))
                samp, cutoff = int(rng.integers(1, 5)), int(rng.integers(0, 6))
                density = float(rng.uniform(0.05, 0.3)) if rng.integers(0, 3) == 0 else None
                if density is not None:
                    sizes = [max(8, n - n % 4) if rng.integers(0, 4) else n for n in sizes]   # (mostly on the 16-byte path: packed)
                cfg = dict(kind="feedback", kinds=kinds, sizes=sizes, m=m, k=k, fb=sorted(pairs), T=T, samp=samp, cutoff=cutoff, density=density)
                x = (rng.standard_normal((T, m if m else sizes[0])) * 0.5 + (1.5 if KINDS[kinds[0]].get("spike") else 0.0)).astype(np.float32)
                res = []
                for where in wheres:
                    with contextlib.redirect_stdout(io.StringIO()):
                        net, rec_var = build_feedback(where, 1000 * seed + case, kinds, sizes, m, k, sorted(pairs), density=density)
                        net.compile()
                        planned = net._coupled_plan([]) is not None
                        o1 = net.run(x[:T // 2], sampling_steps=samp, cutoff=cutoff, verbose=False, enable_grad=False,
                                     record_vars=[(rec_var[0], rec_var[1], False)])
                        o2 = net.run(x[T // 2:], sampling_steps=1, verbose=False, enable_grad=False)
                    res.append((o1.to_numpy("out"), o2.to_numpy("out"), o1.to_numpy(rec_var),
                                {q: v.detach().cpu().numpy() for q, v in net.state.items()}, planned))
                exact = m == 0 and k == 0
                spiking = any(KINDS[q].get("spike") for q in kinds)
                tol = 5e-4 if spiking else 2e-5
                ok = res[0][4] == res[1][4]
                for a, b in zip(res[0][:3], res[1][:3]):
                    ok = ok and (bits_equal(a, b) if exact else close(a, b, tol))
                for q in res[0][3]:
                    ok = ok and (bits_equal(res[0][3][q], res[1][3][q]) if exact else close(res[0][3][q], res[1][3][q], tol))
                cfg["coupled_plan"] = res[0][4]
            else:  # ---- run_batch ----
                n = int(rng.choice([rng.integers(2, 20) * 4, rng.integers(64, 80) * 4, rng.integers(130, max(131, max_n // 4)) * 4]))
                B = int(rng.choice([1, 2, 3, 4, 5, 8, 11, 16, 23, 32, 35]))
                m, k = int(rng.integers(0, 3)), int(rng.integers(0, 3))
                T = int(rng.integers(10, 50))
                samp, cutoff = int(rng.integers(1, 6)), int(rng.integers(0, 6))
                exact_flag = bool(rng.integers(0, 2))
                cfg = dict(kind="batch", n=n, B=B, m=m, k=k, T=T, samp=samp, cutoff=cutoff, exact=exact_flag)
                x = rng.standard_normal((B, T, m if m else 1)).astype(np.float32)
                res = []
                for where in wheres:
                    r2 = np.random.default_rng(1000 * seed + case)
                    with contextlib.redirect_stdout(io.StringIO()):
                        net = Network(1e-2, **where)
                        if m:
                            net.add_func_node("inp", m, "identity")
                        add_population(net, "rnn", "tanh", n, r2)
                        if k:
                            net.add_func_node("out", k, "identity")
                            net.add_edge("rnn", "out", weights=r2.standard_normal((k, n)) / np.sqrt(n))
                        if m:
                            net.add_edge("inp", "rnn", weights=r2.standard_normal((n, m)))
                        obs = net.run_batch(x, sampling_steps=samp, cutoff=cutoff, exact=exact_flag)
                    res.append(([o.to_numpy("out") for o in obs], [o["state"].cpu().numpy() for o in obs]))
                ok = True
                for b in range(B):
                    # read-outs are torch products on the device: bits only for the population's own output
                    same = bits_equal if (k == 0) else (lambda u, v: close(u, v, 2e-5))
                    ok = ok and same(res[0][0][b], res[1][0][b]) and bits_equal(res[0][1][b], res[1][1][b])
        except Exception as e:  # noqa: BLE001
            ok = False
            cfg["exception"] = f"{type(e).__name__}: {e}"
        if ok and os.environ.get("FUZZ_VERBOSE") and cfg.get("kind") in ("fit", "variants"):
            print("ok", json.dumps({q: v for q, v in cfg.items() if q in ("kind", "kinds", "sizes", "edges", "varied", "fused_adjoint", "coupled_plan")}), flush=True)
        if not ok:
            fails += 1
            print(json.dumps(dict(case=case, seed=seed, **{q: (v if not isinstance(v, (np.integer,)) else int(v)) for q, v in cfg.items()})), flush=True)
    print(f"fuzz_coupled_hip_vs_oracle: {cases} cases, seed {seed}: {fails} failures", flush=True)
    return fails


if __name__ == "__main__":
    a = sys.argv[1:]
    sys.exit(1 if main(int(a[0]) if a else 40, int(a[1]) if len(a) > 1 else 0, int(a[2]) if len(a) > 2 else 1100) else 0)
