"""Differential fuzz, build container only: random chains  inp --edge--> rnn --edge--> out  with random edge variants
(plain, masked, delay, filter, delay + filter) on both edges, random sampling / cutoff, two consecutive runs (edge state
carried over) -- the reference's rectipy.Network walking the graph step by step against rectipy_amd.Network on its
fused-window plan (edges evaluated per window, CPU oracle arithmetic).  Prints the worst deviation."""
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


def edge_kwargs(rng, n_in, n_out, kind):
    if kind == "plain":
        return {}
    if kind == "masked":
        return dict(mask=(rng.uniform(size=(n_out, n_in)) > 0.4).astype(np.float64))
    delays = rng.integers(0, int(rng.integers(1, 5)), n_in)
    F = rng.standard_normal((n_in, n_in)) * 0.2 / max(1.0, np.sqrt(n_in))
    return {"memory": dict(delays=delays), "filter": dict(filter_weights=F),
            "memfilter": dict(delays=delays, filter_weights=F)}[kind]


def main(cases=40, seed=0):
    rng = np.random.default_rng(seed)
    kinds = ["plain", "masked", "memory", "filter", "memfilter"]
    worst, worst_at = 0.0, None
    for case in range(cases):
        n, m, k = int(rng.integers(4, 28)), int(rng.integers(1, 6)), int(rng.integers(1, 4))
        while m == n or k == n:   # (square edge matrices are transposed by both: keep the shapes unambiguous anyway)
            n += 1
        dt = 1e-2
        P = dict(W=rng.standard_normal((n, n)) * 0.8 / np.sqrt(n), tau=rng.uniform(5, 10, n), k=rng.uniform(0.5, 1.5, n),
                 eta=rng.standard_normal(n) * 0.1, y0=rng.standard_normal(n) * 0.5)
        W_in, W_out = rng.standard_normal((n, m)), rng.standard_normal((k, n))
        kin, kout = kinds[int(rng.integers(0, 5))], kinds[int(rng.integers(0, 5))]
        kw_in, kw_out = edge_kwargs(rng, m, n, kin), edge_kwargs(rng, n, k, kout)
        ref = RefNetwork(dt, dtype=torch.float64)
        ref.add_func_node("inp", m, "identity")
        ref.add_node("rnn", rate_node(RefRateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, torch.float64),
                     node_type="diff_eq")
        ref.add_func_node("out", k, "identity")
        ref.add_edge("inp", "rnn", weights=W_in.copy(), dtype=torch.float64, **{a: np.array(b) for a, b in kw_in.items()})
        ref.add_edge("rnn", "out", weights=W_out.copy(), dtype=torch.float64, **{a: np.array(b) for a, b in kw_out.items()})
        amd = rectipy_amd.Network(dt, device="cpu", backend=OracleBackend())
        amd.add_func_node("inp", m, "identity")
        amd.add_diffeq_node("rnn", TANH, input_var="li_op/I_ext", output_var="li_op/v", weights=P["W"],
                            source_var="tanh_op/r", target_var="li_op/r_in", dtype=torch.float32,
                            node_vars={"all/li_op/tau": P["tau"], "all/li_op/k": P["k"], "all/li_op/eta": P["eta"],
                                       "all/li_op/v": P["y0"]})
        amd.add_func_node("out", k, "identity")
        amd.add_edge("inp", "rnn", weights=W_in.copy(), **{a: np.array(b) for a, b in kw_in.items()})
        amd.add_edge("rnn", "out", weights=W_out.copy(), **{a: np.array(b) for a, b in kw_out.items()})
        amd.compile()
        assert amd._chain_plan() is not None, "the fused plan must take this graph"
        for part in range(2):
            T = int(rng.integers(2, 16))
            x = rng.standard_normal((T, m)) * 0.5
            kw = dict(sampling_steps=int(rng.integers(1, 4)), cutoff=int(rng.integers(0, 3)), verbose=False, enable_grad=False)
            a = ref.run(torch.tensor(x, dtype=torch.float64), **kw).to_numpy("out")
            b = amd.run(x.astype(np.float32), **kw).to_numpy("out")
            dev = float(np.abs(a - b).max()) if a.size else 0.0
            dev = max(dev, float(np.abs(ref.state["rnn"].detach().numpy() - amd.state["rnn"].detach().numpy()).max()))
            scale = max(1.0, float(np.abs(a).max()) if a.size else 1.0)
            if dev / scale > worst:
                worst, worst_at = dev / scale, (case, kin, kout, n, m, k, part)
            if dev / scale > 5e-5:
                print("MISMATCH", dev, case, kin, kout, n, m, k, part, kw)
                return 1
    print(f"{cases} random chains with edge variants: worst relative deviation {worst:.3e} at {worst_at}")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
