"""Differential fuzz, build container only: random SEQUENCES of read-out training calls -- fit_rls on one array (random
update / sampling strides; the reference's fit_rls records node dictionaries for `record_vars`, so none are asked for), fit_rls over epochs, run, reset, test(), and a closing fit_ridge (which
adds the read-out node) followed by a run through the new node -- on the reference's rectipy.Network around its own
RateNet (float64) and on rectipy_amd.Network with the CPU oracle backend (float32).  Compared after every call: the
recorded outputs and losses, the RLS edge's weights and P, the state, the ridge solution.  Prints the worst deviation
relative to the scale of the compared quantity."""
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
from oracle.backend import OracleBackend  # noqa: E402
import rectipy_amd  # noqa: E402

TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"


def build(rng, n, m, k, dt, beta, alpha, blocked=False):
    P = dict(W=rng.standard_normal((n, n)) * 0.9 / np.sqrt(n), tau=rng.uniform(0.5, 2.0, n), k=rng.uniform(0.5, 1.5, n),
             eta=rng.standard_normal(n) * 0.1, y0=rng.standard_normal(n) * 0.5)
    W_in = rng.standard_normal((n, m))
    W0 = rng.standard_normal((k, n)) * 0.1
    ref = RefNetwork(dt, dtype=torch.float64)
    ref.add_func_node("inp", m, "identity")
    ref.add_node("rnn", rate_node(RefRateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, torch.float64),
                 node_type="diff_eq", op="li_op")
    ref.add_func_node("out", k, "identity")
    ref.add_edge("inp", "rnn", weights=W_in.copy(), dtype=torch.float64)
    ref.add_edge("rnn", "out", weights=W0.copy(), train="rls", beta=beta, alpha=alpha, dtype=torch.float64)
    amd = rectipy_amd.Network(dt, device="cpu", backend=OracleBackend())
    amd.add_func_node("inp", m, "identity")
    amd.add_diffeq_node("rnn", TANH, input_var="li_op/I_ext", output_var="li_op/v", weights=P["W"], source_var="tanh_op/r",
                        target_var="li_op/r_in", dtype=torch.float32,
                        node_vars={"all/li_op/tau": P["tau"], "all/li_op/k": P["k"], "all/li_op/eta": P["eta"],
                                   "all/li_op/v": P["y0"]})
    amd.add_func_node("out", k, "identity")
    amd.add_edge("inp", "rnn", weights=W_in.copy())
    edge = amd.add_edge("rnn", "out", weights=W0.copy(), train="rls", beta=beta, alpha=alpha)
    if blocked:  # the blocked symmetric sweep (the default from n_in = 8192) forced at this size
        edge.tiled, edge.symmetric = True, True
    return ref, amd


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return float("inf")
    return float(np.abs(a - b).max() / max(1.0, np.abs(a).max())) if a.size else 0.0


def main(cases=30, seed=0, tol=2e-3, blocked=False):
    rng = np.random.default_rng(seed)
    worst, worst_at = 0.0, None
    quiet = contextlib.redirect_stdout(io.StringIO())
    for case in range(cases):
        n, m, k = int(rng.integers(4, 40)), int(rng.integers(1, 4)), int(rng.integers(1, 4))
        if blocked:
            n = 4 * ((n + 3) // 4)   # (the symmetric pass is a 16-byte path)
        dt, beta, alpha = 1e-1, float(rng.uniform(0.9, 1.0)), float(rng.uniform(0.5, 3.0))
        ref, amd = build(rng, n, m, k, dt, beta, alpha, blocked)
        log = []
        ops = [["fit_rls", "fit_rls_epochs", "run", "reset", "test"][int(rng.integers(0, 5))] for _ in range(int(rng.integers(2, 6)))]
        for op in ops + ["fit_ridge"]:
            log.append(op)
            dev = 0.0
            T = int(rng.integers(3, 40))
            x, tgt = rng.standard_normal((T, m)) * 0.5, rng.standard_normal((T, k))
            x32, tgt32 = x.astype(np.float32), tgt.astype(np.float32)
            with quiet:
                if op == "fit_rls":
                    kw = dict(update_steps=int(rng.integers(1, 4)), sampling_steps=int(rng.integers(1, 4)), verbose=False)
                    a, b = ref.fit_rls(x, tgt, **kw), amd.fit_rls(x32, tgt32, **kw)
                    dev = max(rel(a.to_numpy("out"), b.to_numpy("out")),
                              rel([float(l) for l in a["loss"]], [float(l) for l in b["loss"]]))
                elif op == "fit_rls_epochs":
                    E = int(rng.integers(1, 4))
                    xs = [rng.standard_normal((T, m)) * 0.5 for _ in range(E)]
                    ts = [rng.standard_normal((T, k)) for _ in range(E)]
                    kw = dict(update_steps=int(rng.integers(1, 3)), verbose=False)
                    ref.fit_rls(xs, ts, **kw)
                    amd.fit_rls([v.astype(np.float32) for v in xs], [v.astype(np.float32) for v in ts], **kw)
                elif op == "run":
                    kw = dict(sampling_steps=int(rng.integers(1, 4)), cutoff=int(rng.integers(0, 3)), verbose=False,
                              enable_grad=bool(rng.integers(0, 2)))
                    dev = rel(ref.run(x, **kw).to_numpy("out"), amd.run(x32, **kw).to_numpy("out"))
                elif op == "reset":
                    y = rng.standard_normal(n) * 0.5
                    ref.reset({"rnn": y})
                    amd.reset({"rnn": y})
                elif op == "test":
                    kw = dict(loss=["mse", "l1"][int(rng.integers(0, 2))], sampling_steps=1, verbose=False)
                    (a, la), (b, lb) = ref.test(x, tgt, **kw), amd.test(x32, tgt32, **kw)
                    dev = max(rel(a.to_numpy("out"), b.to_numpy("out")), rel(la, lb))
                else:
                    T = int(rng.integers(n + 5, 3 * n + 10))
                    x, tgt = rng.standard_normal((T, m)) * 0.5, rng.standard_normal((T, k))
                    kw = dict(sampling_steps=1, alpha=float(rng.uniform(1e-3, 1e-1)), verbose=False)
                    a = ref.fit_ridge(x, tgt, **kw)
                    b = amd.fit_ridge(x.astype(np.float32), tgt.astype(np.float32), **kw)
                    dev = max(rel(a["w_out"].detach().numpy(), b["w_out"].detach().cpu().numpy()),
                              rel(a["y"].detach().numpy(), b["y"].detach().cpu().numpy()))
                    x2 = rng.standard_normal((7, m)) * 0.5
                    dev = max(dev, rel(ref.run(x2, sampling_steps=1, verbose=False, enable_grad=False).to_numpy("out"),
                                       amd.run(x2.astype(np.float32), sampling_steps=1, verbose=False,
                                               enable_grad=False).to_numpy("out")))
            er, ea = ref.get_edge("rnn", "out"), amd.get_edge("rnn", "out")
            dev = max(dev, rel(er.weights.detach().numpy(), ea.weights.detach().cpu().numpy()),
                      rel(er.P.detach().numpy(), ea.P.detach().cpu().numpy()),
                      rel(ref.state["rnn"].detach().numpy(), amd.state["rnn"].detach().cpu().numpy()))
            if dev > worst:
                worst, worst_at = dev, (case, list(log), n, m, k)
            if not dev <= tol:
                print("MISMATCH", dev, case, log, n, m, k)
                return 1
    print(f"{cases} random read-out training sequences: worst relative deviation {worst:.3e} at {worst_at}")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
