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
from make_golden import import_reference, rate_node  # noqa: E402

RefNetwork, RefRateNet, _, _ = import_reference()
from rectipy.network import FeedbackNetwork as RefFeedbackNetwork  # noqa: E402
from oracle.backend import OracleBackend  # noqa: E402
import rectipy_amd  # noqa: E402
from rectipy_amd.feedback import FeedbackNetwork  # noqa: E402

TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"


def pop(rng, n):
    return dict(W=rng.standard_normal((n, n)) * 0.8 / np.sqrt(n), tau=rng.uniform(0.5, 2.0, n), k=rng.uniform(0.5, 1.5, n),
                eta=rng.standard_normal(n) * 0.1, y0=rng.standard_normal(n) * 0.5)


def build(rng, sizes, m, k, feedback, dt, trainable=False):
    labels = ["a", "b", "c"][:len(sizes)]
    pops = [pop(rng, n) for n in sizes]
    choices = [None, ["weights"], ["li_op/tau", "li_op/eta"], ["weights", "li_op/k"]]
    trains = [choices[int(rng.integers(0, 4))] if trainable else None for _ in sizes]
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
    for lab, P, train in zip(labels, pops, trains):
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
            ref.add_edge(s, t, weights=W.copy(), feedback=True, dtype=torch.float64)
            amd.add_edge(s, t, weights=W.copy(), feedback=True)
            fb.append((s, t))
    return ref, amd, labels, (m or sizes[0]), fb, list(zip(order[:-1], order[1:])), trains


def params_dev(ref, amd, labels, edges, trains):
    dev = 0.0
    for s, t in edges:
        dev = max(dev, float(np.abs(ref.get_edge(s, t).weights.detach().numpy() - amd.get_edge(s, t).weights.detach().numpy()).max()))
    for lab, train in zip(labels, trains):
        for name in train or []:
            a = ref.get_node(lab)._args[{"weights": 1, "li_op/tau": 2, "li_op/k": 3, "li_op/eta": 4}[name]]
            dev = max(dev, float(np.abs(a.detach().numpy() - amd.get_node(lab)[name].detach().numpy()).max()))
    return dev


def main(cases=30, seed=0, tol=5e-5, trainable=False):
    rng = np.random.default_rng(seed)
    worst, worst_at = 0.0, None
    for case in range(cases):
        sizes = [int(rng.integers(3, 20)) for _ in range(int(rng.integers(2, 4)))]
        m, k = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        feedback = bool(rng.integers(0, 2))
        dt = 5e-2
        feedback = feedback and not trainable
        ref, amd, labels, width, fb, edges, trains = build(rng, sizes, m, k, feedback, dt, trainable)
        n_out = k or sizes[-1]
        log = []
        for _ in range(int(rng.integers(2, 6))):
            op = ["run", "run_nograd", "reset", "reset_one", "forward", "fit"][int(rng.integers(0, 6 if trainable else 5))]
            log.append(op)
            dev = 0.0
            if op in ("run", "run_nograd"):
                T = int(rng.integers(2, 25))
                x = rng.standard_normal((T, width)) * 0.5
                kw = dict(sampling_steps=int(rng.integers(1, 4)), cutoff=int(rng.integers(0, 3)), verbose=False,
                          enable_grad=op == "run")
                a = ref.run(torch.tensor(x), **kw).to_numpy("out")
                b = amd.run(x.astype(np.float32), **kw).to_numpy("out")
                dev = float(np.abs(a - b).max()) if a.size and a.shape == b.shape else (0.0 if a.shape == b.shape else np.inf)
            elif op == "reset":
                ys = {lab: rng.standard_normal(n) * 0.5 for lab, n in zip(labels, sizes)}
                ref.reset(ys)
                amd.reset(ys)
            elif op == "reset_one":
                i = int(rng.integers(0, len(labels)))
                ys = {labels[i]: rng.standard_normal(sizes[i]) * 0.5}
                ref.reset(ys)
                amd.reset(ys)
            elif op == "fit":
                w = int(rng.integers(2, 9))
                T = w * int(rng.integers(1, 4)) + int(rng.integers(0, 2))
                x, tgt = rng.standard_normal((T, width)) * 0.5, rng.standard_normal((T, n_out)) * 0.3
                kw = dict(optimizer=["sgd", "adam"][int(rng.integers(0, 2))], lr=1e-2, update_steps=w,
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
                    x = rng.standard_normal(width) * 0.5
                    with torch.no_grad():
                        a = ref.forward(torch.tensor(x)).detach().numpy()
                        b = amd.forward(torch.tensor(x, dtype=torch.float32)).detach().numpy()
                    dev = max(dev, float(np.abs(a - b).max()))
            for lab in labels:
                dev = max(dev, float(np.abs(ref.state[lab].detach().numpy() - amd.state[lab].detach().numpy()).max()))
            if dev > worst:
                worst, worst_at = dev, (case, list(log), sizes, m, k, fb)
            if not dev <= tol:
                print("MISMATCH", dev, case, log, sizes, m, k, fb)
                return 1
    print(f"{cases} random population chains: worst deviation {worst:.3e} at {worst_at}")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                  trainable=len(sys.argv) > 3 and sys.argv[3] == "train", tol=2e-3 if len(sys.argv) > 3 else 5e-5))
