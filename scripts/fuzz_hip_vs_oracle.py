"""GPU box: the package on the HIP library against the package on the CPU oracle, through the front end, on RANDOM shapes
and call sequences -- every node template, N from 5 to ~2100 (ragged chunks, 4-byte paths, several row groups), random
input width / read-out width, run (sampling, cutoff, recorded variables, truncate_steps) / reset / forward() steps /
fit_bptt windows (sgd / adam; the templates with an adjoint kernel) / fit_rls / fit_ridge.  Forward quantities must be
BIT-EQUAL (same summation order); quantities behind a torch reduction on the device (read-outs, losses,
optimiser steps) are compared to 2e-5 relative.  Prints one JSON line per failure and a summary.

    python scripts/fuzz_hip_vs_oracle.py [cases] [seed] [max_n]
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
from rectipy_amd import Network  # noqa: E402

DEV = os.environ.get("FUZZ_DEVICE", "cuda:0")  # ("cpu": the harness against itself, no GPU needed)
P = "neuron_model_templates."
MODELS = {
    "tanh": dict(path=P + "rate_neurons.leaky_integrator.tanh", op="li_op", src="tanh_op/r", tgt="li_op/r_in", out=["li_op/v", "tanh_op/r"],
                 train=[None, ["weights"], ["li_op/tau", "li_op/eta"], ["weights", "li_op/k"]], dt=1e-2, drive=0.0),
    "sigmoid": dict(path=P + "rate_neurons.leaky_integrator.sigmoid", op="li_op", src="sigmoid_op/r", tgt="li_op/r_in", out=["li_op/v"],
                    train=[None, ["weights"], ["sigmoid_op/s", "li_op/eta"]], dt=1e-2, drive=0.0),
    "qif": dict(path=P + "spiking_neurons.qif.qif", op="qif_op", src="qif_op/s", tgt="qif_op/s_in", out=["qif_op/s"], spike=True,
                train=[None, ["weights"], ["qif_op/eta"]], dt=5e-3, drive=2.0, theta=1.0, reset=-1.0,
                vars={"tau": (0.8, 1.2), "k": (0.5, 1.5), "eta": (0.5, 3.0), "tau_s": (0.5, 2.0)}, wscale=2.0),
    "qif_sfa": dict(path=P + "spiking_neurons.qif.qif_sfa", op="qif_sfa_op", src="qif_sfa_op/s", tgt="qif_sfa_op/s_in", out=["qif_sfa_op/s"],
                    spike=True, train=[None], dt=5e-3, drive=2.0, theta=1.0, reset=-1.0,
                    vars={"tau": (0.8, 1.2), "k": (0.5, 1.5), "eta": (0.5, 3.0), "tau_s": (0.5, 2.0)}, wscale=2.0),
    "lif": dict(path=P + "spiking_neurons.lif.lif", op="lif_op", src="lif_op/s", tgt="lif_op/s_in", out=["lif_op/s"], spike=True,
                train=[None, ["weights"]], dt=1e-2, drive=3.0, theta=1.0, reset=0.0,
                vars={"tau": (0.4, 0.8), "k": (0.5, 1.5), "eta": (2.0, 5.0), "tau_s": (0.2, 0.6)}, wscale=1.0),
    "ik": dict(path=P + "spiking_neurons.ik.ik", op="ik_op", src="ik_op/s", tgt="ik_op/s_in", out=["ik_op/s"], spike=True,
               train=[None], dt=5e-2, drive=80.0, theta=40.0, reset=-60.0, vars={}, wscale=5.0),
    "iku": dict(path=P + "spiking_neurons.ik.iku", op="iku_op", src="iku_op/s", tgt="iku_op/s_in", out=["iku_op/s"], spike=True,
                train=[None], dt=5e-2, drive=80.0, theta=40.0, reset=-60.0, vars={}, wscale=5.0),
    "ik_biexp": dict(path=P + "spiking_neurons.ik.ik_biexp", op="ik_biexp_op", src="ik_biexp_op/s", tgt="ik_biexp_op/s_in",
                     out=["ik_biexp_op/s"], spike=True, train=[None], dt=5e-2, drive=80.0, theta=40.0, reset=-60.0, vars={}, wscale=5.0),
}


def build(where, seed, name, n, m, k, train, out_var, readout, sparse=0.0):
    M = MODELS[name]
    rng = np.random.default_rng(seed)
    net = Network(M["dt"], **where)
    if m:
        net.add_func_node("inp", m, "identity")
    node_vars = {}
    if name in ("tanh", "sigmoid"):
        W = rng.standard_normal((n, n)) * 1.2 / np.sqrt(n)
        node_vars = {"all/li_op/tau": rng.uniform(0.5, 2.0, n), "all/li_op/k": rng.uniform(0.5, 1.5, n),
                     "all/li_op/eta": rng.standard_normal(n) * 0.1, "all/li_op/v": rng.standard_normal(n) * 0.5}
        kw = {}
    else:
        W = np.abs(rng.standard_normal((n, n))) * M["wscale"] / n
        op = M["op"]
        node_vars = {f"all/{op}/{q}": rng.uniform(lo, hi, n) for q, (lo, hi) in M["vars"].items()}
        kw = dict(spike_var=f"{op}/spike", reset_var=f"{op}/v", spike_threshold=M["theta"], spike_reset=M["reset"])
    if sparse:  # a fixed sparse W is stored packed ("auto": when W is not trained) -- the packed step kernels
        W = W * (rng.uniform(size=W.shape) < sparse) / sparse
        kw["sparse_weights"] = "auto"
    net.add_diffeq_node("rnn", M["path"], weights=W, source_var=M["src"], target_var=M["tgt"], input_var=f"{M['op']}/I_ext",
                        output_var=out_var, dtype=torch.float32, train_params=train, node_vars=node_vars, **kw)
    if k:
        net.add_func_node("out", k, ["identity", "tanh"][int(rng.integers(0, 2))])
        net.add_edge("rnn", "out", weights=rng.standard_normal((k, n)) / np.sqrt(n), **readout)
    if m:
        net.add_edge("inp", "rnn", weights=rng.standard_normal((n, m)))
    return net


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and bool(((a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))).all())


def close(a, b, tol=2e-5):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and (a.size == 0 or bool(np.abs(a - b).max() <= tol * max(1.0, np.abs(a).max())))


def main(cases=50, seed=0, max_n=2100):
    rng = np.random.default_rng(seed)
    quiet = contextlib.redirect_stdout(io.StringIO())
    fails = 0
    sizes = [5, 8, 63, 64, 65, 250, 256, 257, 511, 1000, 1024, 1028, 2052]
    for case in range(cases):
        name = list(MODELS)[int(rng.integers(0, len(MODELS)))]
        M = MODELS[name]
        n = int(sizes[int(rng.integers(0, len(sizes)))] if rng.integers(0, 2) else rng.integers(5, max_n))
        n = min(n, max_n)
        m, k = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        train = M["train"][int(rng.integers(0, len(M["train"])))]
        out_var = M["out"][int(rng.integers(0, len(M["out"])))]
        mode = ["plain", "plain", "gd", "rls"][int(rng.integers(0, 4))] if k else "plain"
        readout = {"gd": dict(train="gd"), "rls": dict(train="rls", beta=0.98, alpha=1.5)}.get(mode, {})
        if mode == "rls":
            train = None
        sparse = float(rng.uniform(0.03, 0.3)) if (rng.integers(0, 3) == 0 and not (train and "weights" in train)) else 0.0
        if sparse:
            n = max(8, n - n % 4)  # (the packed layout needs n % 4 == 0)
        args = (int(rng.integers(0, 1 << 30)), name, n, m, k, train, out_var, readout, sparse)
        cpu = build(dict(device="cpu", backend=OracleBackend()), *args)
        gpu = build(dict(device=DEV) if DEV != "cpu" else dict(device="cpu", backend=OracleBackend()), *args)
        if mode == "rls" and n % 4 == 0 and rng.integers(0, 2):
            # the symmetric pass and its blocked sweep (chosen by size from n_in = 8192) forced at this size
            block, rows = int([1, 2, 4, 8][int(rng.integers(0, 4))]), int([64, 128][int(rng.integers(0, 2))])
            for net_ in (cpu, gpu):
                e_ = net_.get_edge("rnn", "out")
                e_.tiled, e_.symmetric, e_.block, e_.BLOCK_ROWS = True, True, block, rows
        width = m or n
        n_out = k or n
        # an input node's projection is a torch product on the device (rocBLAS) and on the host (MKL): equal to rounding only,
        # so bit-equality is asked for where the population is driven directly (m == 0); behind a projection a spiking
        # population may move a spike by a step, hence the loose bound there
        log, trained = [], m > 0
        loose = 2e-3 if (m > 0 and M.get("spike")) else 5e-5
        for _ in range(int(rng.integers(2, 6))):
            ops = ["run", "run_nograd", "reset", "forward"]
            if train or mode == "gd":
                ops.append("fit")
            if mode == "rls":
                ops.append("fit_rls")
            if k == 0 and not M.get("spike"):
                ops.append("fit_ridge")
            if not M.get("spike") and n % 4 == 0 and not sparse:
                ops.append("run_batch")
            op = ops[int(rng.integers(0, len(ops)))]
            log.append(op)
            ok = True
            T = int(rng.integers(20, 90)) if M.get("spike") else int(rng.integers(2, 40))
            x = (rng.standard_normal((T, width)) * 0.5 * max(1.0, M["drive"] / 4) + M["drive"]).astype(np.float32)
            with quiet:
                if op in ("run", "run_nograd"):
                    kw = dict(sampling_steps=int(rng.integers(1, 5)), cutoff=int(rng.integers(0, 4)), verbose=False,
                              enable_grad=op == "run")
                    if rng.integers(0, 3) == 0:
                        kw["truncate_steps"] = int(rng.integers(2, 10))
                    rec = rng.integers(0, 2) == 0
                    if rec:
                        kw["record_vars"] = [("rnn", f"{M['op']}/v", False)]
                    a, b = cpu.run(x, **kw), gpu.run(x, **kw)
                    exact = k == 0 and not trained
                    ok = bits_equal(a.to_numpy("out"), b.to_numpy("out")) if exact else close(a.to_numpy("out"), b.to_numpy("out"), loose)
                    if rec:
                        va, vb = a.to_numpy(("rnn", f"{M['op']}/v")), b.to_numpy(("rnn", f"{M['op']}/v"))
                        ok = ok and (bits_equal(va, vb) if not trained else close(va, vb, loose if not M.get("spike") else 1.0))
                elif op == "run_batch":
                    B = int([2, 3, 5, 8, 9, 16, 17, 32, 33][int(rng.integers(0, 9))])
                    Tb = int(rng.integers(2, 12))
                    xb = (rng.standard_normal((B, Tb, width)) * 0.5).astype(np.float32)
                    y0 = (rng.standard_normal((B, n)) * 0.5).astype(np.float32)
                    kw = dict(sampling_steps=int(rng.integers(1, 4)), cutoff=int(rng.integers(0, 2)), y0=y0)
                    oa, ob = cpu.run_batch(xb, **kw), gpu.run_batch(xb, **kw)
                    for a, b in zip(oa, ob):
                        same = bits_equal if (k == 0 and m == 0 and not trained) else (lambda u, v: close(u, v, loose))
                        ok = ok and same(a.to_numpy("out"), b.to_numpy("out")) and \
                            (bits_equal if m == 0 and not trained else (lambda u, v: close(u, v, loose)))(a["state"].cpu().numpy(), b["state"].cpu().numpy())
                elif op == "reset":
                    y = rng.uniform(-0.5, 0.5, cpu.state["rnn"].shape[0]).astype(np.float32)
                    if name.startswith("ik"):
                        y = None
                    cpu.reset(None if y is None else {"rnn": y})
                    gpu.reset(None if y is None else {"rnn": y})
                elif op == "forward":
                    cpu.compile()
                    gpu.compile()
                    for t in range(min(T, 4)):
                        with torch.no_grad():
                            a = cpu.forward(torch.tensor(x[t])).numpy()
                            b = gpu.forward(torch.tensor(x[t], device=DEV)).cpu().numpy()
                        ok = ok and (bits_equal(a, b) if (k == 0 and not trained) else close(a, b, loose))
                elif op == "fit":
                    w = int(rng.integers(2, 12))
                    Tf = w * int(rng.integers(1, 3))
                    xf = x[:Tf] if Tf <= T else np.concatenate([x] * (Tf // T + 1))[:Tf]
                    tgt = (rng.standard_normal((Tf, n_out)) * 0.2).astype(np.float32)
                    kw = dict(optimizer=["sgd", "adam"][int(rng.integers(0, 2))], lr=1e-3, update_steps=w, sampling_steps=1,
                              verbose=False)
                    a, b = cpu.fit_bptt(xf, tgt, **kw), gpu.fit_bptt(xf, tgt, **kw)
                    trained = True
                    ok = close(a.to_numpy("out"), b.to_numpy("out"), loose)
                    for p in train or []:
                        ok = ok and close(cpu.get_node("rnn")[p].detach().numpy(), gpu.get_node("rnn")[p].detach().cpu().numpy(), loose)
                elif op == "fit_rls":
                    tgt = (rng.standard_normal((T, k))).astype(np.float32)
                    kw = dict(update_steps=int(rng.integers(1, 4)), sampling_steps=1, verbose=False)
                    a, b = cpu.fit_rls(x, tgt, **kw), gpu.fit_rls(x, tgt, **kw)
                    ea, eb = cpu.get_edge("rnn", "out"), gpu.get_edge("rnn", "out")
                    same = bits_equal if m == 0 else (lambda u, v: close(u, v, 1e-3))
                    ok = same(ea.P.numpy(), eb.P.cpu().numpy()) and same(ea.weights.numpy(), eb.weights.cpu().numpy()) \
                        and close(a.to_numpy("out"), b.to_numpy("out"), max(loose, 1e-4))
                else:  # fit_ridge: adds the read-out node, so it closes the sequence
                    Tr = n + 20
                    xr = (rng.standard_normal((Tr, width)) * 0.5).astype(np.float32)
                    tgt = (rng.standard_normal((Tr, 2))).astype(np.float32)
                    a, b = cpu.fit_ridge(xr, tgt, sampling_steps=1, alpha=1e-2, verbose=False), \
                        gpu.fit_ridge(xr, tgt, sampling_steps=1, alpha=1e-2, verbose=False)
                    # (T = n + 20 samples: an ill-conditioned solve -- behind a trained W, which differs by rounding between the two
                    # devices, the read-out moves by more than behind identical states: docs/semantics.md, G39)
                    ok = close(a["w_out"].detach().numpy(), b["w_out"].detach().cpu().numpy(), 5e-3 if trained else 2e-4)
            sa, sb = cpu.state["rnn"].detach().numpy(), gpu.state["rnn"].detach().cpu().numpy()
            if M.get("spike") and trained:
                lo = 2 * n if name.startswith("ik") else n  # compare s: v (and IK's u) jump at a spike, which may land a step apart
                sa, sb = sa[lo:lo + n], sb[lo:lo + n]
            ok = ok and (bits_equal(sa, sb) if not trained else close(sa, sb, loose))
            if not ok:
                fails += 1
                print(json.dumps({"fail": True, "case": case, "model": name, "n": n, "m": m, "k": k, "train": train, "out": out_var,
                                  "readout": mode, "sparse": round(sparse, 3), "log": log}), flush=True)
                break
            if op == "fit_ridge":
                break
        del cpu, gpu
        if case % 10 == 9:
            if DEV != "cpu":
                torch.cuda.empty_cache()
            print(json.dumps({"progress": case + 1, "fails": fails}), flush=True)
    print(json.dumps({"cases": cases, "seed": seed, "fails": fails}), flush=True)
    return int(fails > 0)


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    sys.exit(main(*a))
