"""Differential fuzz, build container only (needs /root/reference): random SEQUENCES of Network calls -- run with / without
gradient tracking, in-place state writes, set_var, reset, detach, fit_bptt windows -- on the reference's rectipy.Network
around its own RateNet (hand-written tanh field, float64) and on rectipy_amd.Network with the CPU oracle backend
(float32).  After every call the outputs, the state and the trained parameters are compared.  Prints the worst deviation."""
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


def build(rng, n, k, train, dt):
    P = dict(W=rng.standard_normal((n, n)) * 0.8 / np.sqrt(n), tau=rng.uniform(5, 10, n), k=rng.uniform(0.5, 1.5, n),
             eta=rng.standard_normal(n) * 0.1, y0=rng.standard_normal(n) * 0.5)
    W_out = rng.standard_normal((k, n)) / np.sqrt(n)
    ref = RefNetwork(dt, dtype=torch.float64)
    ref.add_node("rnn", rate_node(RefRateNet, P["W"], P["tau"], P["k"], P["eta"], P["y0"], dt, torch.float64,
                                  train_params=train), node_type="diff_eq", op="li_op")
    ref.add_func_node("out", k, "identity")
    ref.add_edge("rnn", "out", weights=W_out.copy(), train="gd", dtype=torch.float64)
    amd = rectipy_amd.Network(dt, device="cpu", backend=OracleBackend())
    amd.add_diffeq_node("rnn", TANH, input_var="li_op/I_ext", output_var="li_op/v", weights=P["W"], source_var="tanh_op/r",
                        target_var="li_op/r_in", train_params=train, dtype=torch.float32,
                        node_vars={"all/li_op/tau": P["tau"], "all/li_op/k": P["k"], "all/li_op/eta": P["eta"],
                                   "all/li_op/v": P["y0"]})
    amd.add_func_node("out", k, "identity")
    amd.add_edge("rnn", "out", weights=W_out.copy(), train="gd")
    return ref, amd


TOL = 2e-4  # (adam / rmsprop divide by the root of a squared gradient: float32 gradients near zero move a step by O(lr))


def main(cases=40, seed=0):
    rng = np.random.default_rng(seed)
    worst, worst_at = 0.0, None
    for case in range(cases):
        n, k = int(rng.integers(4, 40)), int(rng.integers(1, 4))
        train = [None, ["weights"], ["weights", "li_op/tau"], ["li_op/eta", "li_op/k"]][int(rng.integers(0, 4))]
        dt = 1e-2
        ref, amd = build(rng, n, k, train, dt)
        log = []
        for op_i in range(int(rng.integers(3, 8))):
            op = ["run", "run_nograd", "write", "set_var", "reset", "fit", "detach", "fit_epochs", "set_param",
                  "swap_readout"][int(rng.integers(0, 10))]
            log.append(op)
            if op in ("run", "run_nograd"):
                T = int(rng.integers(2, 14))
                x = rng.standard_normal((T, n)) * 0.3
                kw = dict(sampling_steps=int(rng.integers(1, 4)), cutoff=int(rng.integers(0, 3)), verbose=False,
                          enable_grad=op == "run")
                if rng.integers(0, 3) == 0:
                    kw["truncate_steps"] = int(rng.integers(1, 6))
                rec = rng.integers(0, 3) == 0
                if rec:  # (the reference relabels nothing for a hand-built node: its var_map holds "li_op/v")
                    kw["record_vars"] = [("rnn", "li_op/v", bool(rng.integers(0, 2)))]
                oa, ob = ref.run(x, **kw), amd.run(x.astype(np.float32), **kw)
                a, b = oa.to_numpy("out"), ob.to_numpy("out")
                dev = float(np.abs(a - b).max()) if a.size else 0.0
                if a.shape != b.shape or list(oa["steps"]) != list(ob["steps"]):
                    dev = float("inf")
                if rec and a.size:
                    ra, rb = oa.to_numpy(("rnn", "li_op/v")), ob.to_numpy(("rnn", "li_op/v"))
                    dev = max(dev, float(np.abs(ra - rb).max()) if ra.shape == rb.shape else float("inf"))
            elif op == "write":
                v = rng.standard_normal(n) * 0.5
                try:
                    ref.state["rnn"][:] = torch.tensor(v, dtype=torch.float64)
                    ref_ok = True
                except RuntimeError:
                    ref_ok = False   # a leaf that requires grad: torch refuses the in-place write
                try:
                    amd.state["rnn"][:] = torch.tensor(v, dtype=torch.float32)
                    amd_ok = True
                except RuntimeError:
                    amd_ok = False
                if ref_ok != amd_ok:  # same refusal on both sides, or bring them back in line under no_grad
                    with torch.no_grad():
                        ref.state["rnn"][:] = torch.tensor(v, dtype=torch.float64)
                        amd.state["rnn"][:] = torch.tensor(v, dtype=torch.float32)
                dev = 0.0
            elif op == "set_var":
                v = rng.standard_normal(n) * 0.5
                try:  # the reference's own set_var: a TRACKED in-place write into a slice of y (network.py:152-176), which
                    ref.set_var("rnn", "li_op/v", torch.tensor(v, dtype=torch.float64))  # cuts the older graph off
                except RuntimeError:  # y is a leaf that requires grad: torch refuses; this package's set_var does not
                    with torch.no_grad():
                        ref.state["rnn"][:] = torch.tensor(v, dtype=torch.float64)
                amd.set_var("rnn", "li_op/v", v)
                dev = 0.0
            elif op == "set_param":  # a parameter that is not trained gets a new value (reference nodes.py:213-230)
                name = "li_op/eta" if not (train and "li_op/eta" in train) else "li_op/tau"
                if train and name in train:
                    continue
                val = rng.uniform(5, 10, n) if name == "li_op/tau" else rng.standard_normal(n) * 0.1
                ref.set_var("rnn", name, torch.tensor(val, dtype=torch.float64))
                amd.set_var("rnn", name, val)
                dev = 0.0
            elif op == "swap_readout":  # pop_edge + add_edge: the graph is recompiled on the next call
                W_new = rng.standard_normal((k, n)) / np.sqrt(n)
                ref.pop_edge("rnn", "out")
                amd.pop_edge("rnn", "out")
                ref.add_edge("rnn", "out", weights=W_new.copy(), train="gd", dtype=torch.float64)
                amd.add_edge("rnn", "out", weights=W_new.copy(), train="gd")
                dev = 0.0
            elif op == "reset":
                v = rng.standard_normal(n) * 0.5
                ref.reset({"rnn": v})
                amd.reset({"rnn": v})
                dev = 0.0
            elif op == "detach":
                ref.detach()
                amd.detach()
                dev = 0.0
            else:
                import contextlib
                import io
                opt = ["sgd", "adam", "rmsprop", "adamw"][int(rng.integers(0, 4))]
                if op == "fit_epochs":
                    E, T = int(rng.integers(1, 4)), int(rng.integers(2, 9))
                    xs = [rng.standard_normal((T, n)) * 0.3 for _ in range(E)]
                    ts = [rng.standard_normal((T, k)) * 0.1 for _ in range(E)]
                    kw = dict(optimizer=opt, lr=1e-2, sampling_steps=1, verbose=False,
                              loss=["mse", "l1"][int(rng.integers(0, 2))])
                    with contextlib.redirect_stdout(io.StringIO()):
                        a = ref.fit_bptt(xs, ts, **kw).to_numpy("out")
                        b = amd.fit_bptt([v.astype(np.float32) for v in xs], [v.astype(np.float32) for v in ts],
                                         **kw).to_numpy("out")
                else:
                    w = int(rng.integers(1, 6))
                    T = w * int(rng.integers(1, 3)) + int(rng.integers(0, 2)) * int(rng.integers(0, w))
                    x, tgt = rng.standard_normal((T, n)) * 0.3, rng.standard_normal((T, k)) * 0.1
                    kw = dict(optimizer=opt, lr=1e-2, update_steps=w, sampling_steps=int(rng.integers(1, 3)), verbose=False)
                    with contextlib.redirect_stdout(io.StringIO()):
                        a = ref.fit_bptt(x, tgt, **kw).to_numpy("out")
                        b = amd.fit_bptt(x.astype(np.float32), tgt.astype(np.float32), **kw).to_numpy("out")
                dev = float(np.abs(a - b).max()) if a.shape == b.shape else float("inf")
                dev = max(dev, float(np.abs(ref.get_edge("rnn", "out").weights.detach().numpy()
                                            - amd.get_edge("rnn", "out").weights.detach().numpy()).max()))
                if train and "weights" in train:
                    wr = ref.get_node("rnn")._args[1].detach().numpy()  # LI_PARAM_MAP["weights"] = 1
                    dev = max(dev, float(np.abs(wr - amd.get_node("rnn")["weights"].detach().numpy()).max()))
            sdev = float(np.abs(ref.state["rnn"].detach().numpy() - amd.state["rnn"].detach().numpy()).max())
            dev = max(dev, sdev)
            if dev > worst:
                worst, worst_at = dev, (case, list(log), train, n)
            if not dev <= TOL:
                print("MISMATCH", dev, case, log, train, n)
                return 1
    print(f"{cases} random call sequences: worst deviation {worst:.3e} at {worst_at}")
    return 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
