"""Differential fuzz, build container only: random CHAINS of two or three rate populations of different sizes (non-square
edges between them, an optional input node and read-out node), with or without feedback edges (FeedbackNetwork; each
node gets at most one feed-forward and one feedback input -- the reference cannot sum several, network.py:1333,1344),
driven by sequences of run (with / without gradient tracking, sampling, cutoff), reset (whole state or one node) and
forward() single steps.  Reference (float64, its own RateNet / Network / FeedbackNetwork) against rectipy_amd (CPU oracle
backend, float32).  Compared after every call: outputs and every population's state."""
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import import_reference, make_qif, rate_node  # noqa: E402

RefNetwork, RefRateNet, RefSpikeResetNet, _ = import_reference()
from rectipy.network import FeedbackNetwork as RefFeedbackNetwork  # noqa: E402
from oracle.backend import OracleBackend  # noqa: E402
import rectipy_amd  # noqa: E402
from rectipy_amd.feedback import FeedbackNetwork  # noqa: E402

TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
QIF = "neuron_model_templates.spiking_neurons.qif.qif"
QIF_PM = {"in": 0, "spike_var": 1, "weights": 2, "qif_op/tau": 3, "qif_op/k": 4, "qif_op/eta": 5, "qif_op/tau_s": 6}
THETA, V_RESET = 1.0, -1.0


def qif_pop(rng, n):
    return dict(W=np.abs(rng.standard_normal((n, n))) * 2.0 / n, tau=rng.uniform(0.8, 1.2, n), k=rng.uniform(0.5, 1.5, n),
                eta=rng.uniform(0.5, 3.0, n), tau_s=rng.uniform(0.5, 2.0, n),
                y0=np.concatenate([rng.uniform(-0.9, 0.9, n), rng.uniform(0, 0.2, n)]))


def add_qif(ref, amd, lab, P, dt, train=None):
    n = P["W"].shape[0]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    args = (t(P["y0"]), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64), t(P["W"]), t(P["tau"]),
            t(P["k"]), t(P["eta"]), t(P["tau_s"]))
    vm = {"out": (n, 2 * n), "reset_var": (0, n), "qif_op/v": (0, n), "qif_op/s": (n, 2 * n)}
    node = RefSpikeResetNet(make_qif(n), args, vm, dict(QIF_PM), spike_threshold=THETA, spike_reset=V_RESET, dt=dt,
                            dtype=torch.float64, device="cpu", train_params=train)
    ref.add_node(lab, node, node_type="diff_eq", op="qif_op")
    amd.add_diffeq_node(lab, QIF, input_var="qif_op/I_ext", output_var="qif_op/s", weights=P["W"], source_var="qif_op/s",
                        target_var="qif_op/s_in", spike_var="qif_op/spike", reset_var="qif_op/v", dtype=torch.float32,
                        spike_threshold=THETA, spike_reset=V_RESET, train_params=train,
                        node_vars={"all/qif_op/tau": P["tau"], "all/qif_op/k": P["k"], "all/qif_op/eta": P["eta"],
                                   "all/qif_op/tau_s": P["tau_s"], "all/qif_op/v": P["y0"][:n], "all/qif_op/s": P["y0"][n:]})


def pop(rng, n):
    return dict(W=rng.standard_normal((n, n)) * 0.8 / np.sqrt(n), tau=rng.uniform(0.5, 2.0, n), k=rng.uniform(0.5, 1.5, n),
                eta=rng.standard_normal(n) * 0.1, y0=rng.standard_normal(n) * 0.5)


def build(rng, sizes, m, k, feedback, dt, trainable=False, spiking=False, train_fb=False):
    labels = ["a", "b", "c"][:len(sizes)]
    kinds = ["qif" if spiking and rng.integers(0, 2) else "rate" for _ in sizes]
    if spiking and "qif" not in kinds:
        kinds[int(rng.integers(0, len(sizes)))] = "qif"
    pops = [pop(rng, n) if kind == "rate" else qif_pop(rng, n) for n, kind in zip(sizes, kinds)]
    choices = [None, ["weights"], ["li_op/tau", "li_op/eta"], ["weights", "li_op/k"]]
    qchoices = [None, ["weights"], ["qif_op/tau", "qif_op/eta"], ["weights", "qif_op/k", "qif_op/tau_s"]]
    trains = [(qchoices if kind == "qif" else choices)[int(rng.integers(0, 4))] if trainable else None for kind in kinds]
    if feedback:
        ref = RefFeedbackNetwork(dt)
        ref.dtype = torch.float64
        amd = FeedbackNetwork(dt, device="cpu", backend=OracleBackend())
    else:
        ref = RefNetwork(dt, dtype=torch.float64)
        amd = rectipy_amd.Network(dt, device="cpu", backend=OracleBackend())
    if m:
        ref.add_func_node("inp", m, "identity")
        amd.add_func_node("inp", m, "identity")
    for lab, P, train, kind in zip(labels, pops, trains, kinds):
        if kind == "qif":
            add_qif(ref, amd, lab, P, dt, train)
            continue
        ref.add_node(lab, rate_node(RefRateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, torch.float64,
                                    train_params=train), node_type="diff_eq", op="li_op")
        amd.add_diffeq_node(lab, TANH, input_var="li_op/I_ext", output_var="li_op/v", weights=P["W"], source_var="tanh_op/r",
                            target_var="li_op/r_in", dtype=torch.float32, train_params=train,
                            node_vars={"all/li_op/tau": P["tau"], "all/li_op/k": P["k"], "all/li_op/eta": P["eta"],
                                       "all/li_op/v": P["y0"]})
    if k:
        ref.add_func_node("out", k, "identity")
        amd.add_func_node("out", k, "identity")
    order = (["inp"] if m else []) + labels + (["out"] if k else [])
    width = dict(zip(labels, sizes), inp=m, out=k)
    for i, (s, t) in enumerate(zip(order[:-1], order[1:])):
        W = rng.standard_normal((width[t], width[s])) * 0.6 / np.sqrt(width[s])
        gd = "gd" if trainable and rng.integers(0, 2) else None
        if trainable and i == len(order) - 2 and not any(trains):
            gd = "gd"  # (the reference's backward() raises when nothing at all is trainable)
        ref.add_edge(s, t, weights=W.copy(), train=gd, dtype=torch.float64)
        amd.add_edge(s, t, weights=W.copy(), train=gd)
    fb = []
    if feedback:
        targets = list(labels)
        rng.shuffle(targets)
        for t in targets[:int(rng.integers(1, len(labels) + 1))]:
            s = labels[int(rng.integers(0, len(labels)))]
            if s == t or (s, t) in zip(order[:-1], order[1:]):
                continue
            W = rng.standard_normal((width[t], width[s])) * 0.6 / np.sqrt(width[s])
            if any(t == t_ for _, t_ in fb):
                continue  # (one feedback input per node: the reference cannot sum several, network.py:1344)
            gd = "gd" if train_fb and rng.integers(0, 2) else None
            ref.add_edge(s, t, weights=W.copy(), feedback=True, train=gd, dtype=torch.float64)
            amd.add_edge(s, t, weights=W.copy(), feedback=True, train=gd)
            fb.append((s, t))
    return ref, amd, labels, (m or sizes[0]), fb, list(zip(order[:-1], order[1:])), trains


def params_dev(ref, amd, labels, edges, trains):
    dev = 0.0
    for s, t in edges:
        dev = max(dev, float(np.abs(ref.get_edge(s, t).weights.detach().numpy() - amd.get_edge(s, t).weights.detach().numpy()).max()))
    li = {"weights": 1, "li_op/tau": 2, "li_op/k": 3, "li_op/eta": 4}
    for lab, train in zip(labels, trains):
        for name in train or []:
            qif = any(q.startswith("qif_op") for q in train) or len(ref.state[lab]) != amd.get_node(lab)._pop.rows
            a = ref.get_node(lab)._args[QIF_PM[name] if qif else li[name]]
            b = amd.get_node(lab)[name].detach().numpy()
            dev = max(dev, float(np.abs(a.detach().numpy() - b).max() / max(1.0, np.abs(b).max())))
    return dev


def main(cases=30, seed=0, tol=5e-5, trainable=False, spiking=False, train_fb=False, spiking_fit=False):
    rng = np.random.default_rng(seed)
    worst, worst_at = 0.0, None
    for case in range(cases):
        sizes = [int(rng.integers(3, 20)) for _ in range(int(rng.integers(2, 4)))]
        m, k = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        feedback = bool(rng.integers(0, 2))
        dt = 5e-2
        feedback = (feedback and not trainable) or train_fb
        dt = 5e-3 if spiking else dt
        ref, amd, labels, width, fb, edges, trains = build(rng, sizes, m, k, feedback, dt, trainable or spiking_fit, spiking, train_fb)
        edges = edges + fb  # (parameter deviations include the feedback edges)
        state_len = {lab: ref.state[lab].shape[0] for lab in labels}  # (a QIF population's state is [v, s])
        n_out = k or sizes[-1]
        log = []
        n_ops = int(rng.integers(2, 6))
        for i_op in range(n_ops):
            op = ["run", "run_nograd", "reset", "reset_one", "forward", "fit"][int(rng.integers(0, 6 if trainable else 5))]
            if spiking_fit:
                # the reference's detach leaves the spike argument of a SpikeResetNet attached: one backward per network --
                # ONE fit window closes the sequence, and nothing before it records a graph
                op = "fit" if i_op == n_ops - 1 else ("run_nograd" if op == "run" else op)
            log.append(op)
            dev = 0.0
            if op in ("run", "run_nograd"):
                T = int(rng.integers(40, 160)) if spiking else int(rng.integers(2, 25))
                x = rng.standard_normal((T, width)) * 0.5 + (2.0 if spiking else 0.0)
                kw = dict(sampling_steps=int(rng.integers(1, 4)), cutoff=int(rng.integers(0, 3)), verbose=False,
                          enable_grad=op == "run")
                a = ref.run(torch.tensor(x), **kw).to_numpy("out")
                b = amd.run(x.astype(np.float32), **kw).to_numpy("out")
                dev = float(np.abs(a - b).max()) if a.size and a.shape == b.shape else (0.0 if a.shape == b.shape else np.inf)
            elif op == "reset":
                ys = {lab: rng.uniform(-0.9, 0.9, state_len[lab]) for lab in labels}
                ref.reset(ys)
                amd.reset(ys)
            elif op == "reset_one":
                i = int(rng.integers(0, len(labels)))
                ys = {labels[i]: rng.uniform(-0.9, 0.9, state_len[labels[i]])}
                ref.reset(ys)
                amd.reset(ys)
            elif op == "fit":
                w = int(rng.integers(2, 9))
                T = w * int(rng.integers(1, 4)) + int(rng.integers(0, 2))
                if spiking_fit:
                    w = T = int(rng.integers(30, 90))
                x, tgt = rng.standard_normal((T, width)) * 0.5 + (2.0 if spiking else 0.0), rng.standard_normal((T, n_out)) * 0.3
                kw = dict(optimizer=["sgd", "adam"][int(rng.integers(0, 2))], lr=1e-3 if spiking_fit else 1e-2, update_steps=w,
                          sampling_steps=int(rng.integers(1, 3)), verbose=False)
                with contextlib.redirect_stdout(io.StringIO()):
                    try:
                        a = ref.fit_bptt(x, tgt, **kw).to_numpy("out")
                    except RuntimeError as err:  # a window too short for anything trainable to reach its predictions
                        try:
                            amd.fit_bptt(x.astype(np.float32), tgt.astype(np.float32), **kw)
                        except RuntimeError as mine:
                            if str(mine) == str(err):
                                break  # both refuse with the same message; the networks are left mid-call
                        print("MISMATCH: only the reference raised", err, case, log, sizes, m, k)
                        return 1
                    b = amd.fit_bptt(x.astype(np.float32), tgt.astype(np.float32), **kw).to_numpy("out")
                dev = float(np.abs(a - b).max()) if a.shape == b.shape else np.inf
                dev = max(dev, params_dev(ref, amd, labels, edges, trains))
            else:
                ref.compile()
                amd.compile()
                for _ in range(int(rng.integers(1, 4))):
                    x = rng.standard_normal(width) * 0.5 + (2.0 if spiking else 0.0)
                    with torch.no_grad():
                        a = ref.forward(torch.tensor(x)).detach().numpy()
                        b = amd.forward(torch.tensor(x, dtype=torch.float32)).detach().numpy()
                    dev = max(dev, float(np.abs(a - b).max()))
            for lab in labels:
                a, b = ref.state[lab].detach().numpy(), amd.state[lab].detach().numpy()
                if a.shape[0] != dict(zip(labels, sizes))[lab]:
                    a, b = a[a.shape[0] // 2:], b[b.shape[0] // 2:]  # QIF: compare s (v jumps by the reset when a spike lands a step apart)
                dev = max(dev, float(np.abs(a - b).max()))
            if dev > worst:
                worst, worst_at = dev, (case, list(log), sizes, m, k, fb)
            if not dev <= tol:
                print("MISMATCH", dev, case, log, sizes, m, k, fb)
                return 1
    print(f"{cases} random population chains: worst deviation {worst:.3e} at {worst_at}")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                  trainable=len(sys.argv) > 3 and sys.argv[3] == "train", spiking=len(sys.argv) > 3 and sys.argv[3] == "spiking",
                  tol=2e-3 if len(sys.argv) > 3 else 5e-5))
