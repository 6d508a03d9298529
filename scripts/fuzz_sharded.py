"""Fuzz of the row-sharded front end against the single-process one (no reference involved; gloo ranks on the CPU, oracle
arithmetic): every rank builds a ShardedNetwork AND the same unsharded Network from the same seeded global arrays, then
both go through a random sequence of run (with / without gradient tracking, sampling, cutoff) / reset / state write /
forward() steps / fit_bptt (sgd, adam) calls.  After every call: read-out outputs (complete on every rank) within 2e-5 of
the unsharded ones (partial read-out sums are all-reduced), this rank's slice of the state bit-equal while nothing has
been trained and within 2e-5 afterwards, trained parameters within 2e-4 relative.

    python scripts/fuzz_sharded.py [cases] [seed] [world]
"""
import contextlib
import io
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TANH = "neuron_model_templates.rate_neurons.leaky_integrator.tanh"
QIF = "neuron_model_templates.spiking_neurons.qif.qif"


def build(cls, where, rng_seed, n, m, k, model, train, gd):
    rng = np.random.default_rng(rng_seed)
    net = cls(1e-2 if model == "tanh" else 5e-3, **where)
    if m:
        net.add_func_node("inp", m, "identity")
    if model == "tanh":
        net.add_diffeq_node("rnn", TANH, weights=rng.standard_normal((n, n)) * 1.2 / np.sqrt(n), source_var="tanh_op/r",
                            target_var="li_op/r_in", input_var="li_op/I_ext", output_var="li_op/v", dtype=torch.float32,
                            train_params=train,
                            node_vars={"all/li_op/tau": rng.uniform(0.5, 2.0, n), "all/li_op/k": rng.uniform(0.5, 1.5, n),
                                       "all/li_op/eta": rng.standard_normal(n) * 0.1, "all/li_op/v": rng.standard_normal(n) * 0.5})
    else:
        net.add_diffeq_node("rnn", QIF, weights=np.abs(rng.standard_normal((n, n))) * 2.0 / n, source_var="qif_op/s",
                            target_var="qif_op/s_in", input_var="qif_op/I_ext", output_var="qif_op/s",
                            spike_var="qif_op/spike", reset_var="qif_op/v", spike_threshold=1.0, spike_reset=-1.0,
                            dtype=torch.float32, train_params=train,
                            node_vars={"all/qif_op/tau": rng.uniform(0.8, 1.2, n), "all/qif_op/k": rng.uniform(0.5, 1.5, n),
                                       "all/qif_op/eta": rng.uniform(0.5, 3.0, n), "all/qif_op/tau_s": rng.uniform(0.5, 2.0, n),
                                       "all/qif_op/v": rng.uniform(-0.9, 0.9, n), "all/qif_op/s": rng.uniform(0, 0.2, n)})
    net.add_func_node("out", k, "identity")
    if m:
        net.add_edge("inp", "rnn", weights=rng.standard_normal((n, m)))
    net.add_edge("rnn", "out", weights=rng.standard_normal((k, n)) / np.sqrt(n), train="gd" if gd else None)
    return net


def worker(rank, port, world, cases, seed, fails):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.backend import OracleBackend
        from rectipy_amd import Network
        from rectipy_amd.sharded import ShardedNetwork, shard_bounds
        rng = np.random.default_rng(seed)
        quiet = contextlib.redirect_stdout(io.StringIO())
        worst = 0.0
        for case in range(cases):
            n = 8 * world * int(rng.integers(1, 6))
            m, k = int(rng.integers(0, 3)), int(rng.integers(1, 4))
            model = ["tanh", "tanh", "qif"][int(rng.integers(0, 3))]
            train = [None, ["weights"], ["weights", "li_op/tau"], ["li_op/eta", "li_op/k"]][int(rng.integers(0, 4))] \
                if model == "tanh" else [None, ["weights"], ["qif_op/eta"]][int(rng.integers(0, 3))]
            gd = bool(rng.integers(0, 2)) or not train
            args = (int(rng.integers(0, 1 << 30)), n, m, k, model, train, gd)
            sh = build(ShardedNetwork, dict(device="cpu", backend=OracleBackend()), *args)
            full = build(Network, dict(device="cpu", backend=OracleBackend()), *args)
            row0, rows = shard_bounds(n, world, rank)
            width = m or n
            trained, log = False, []
            sl_in = (lambda x: x) if m else (lambda x: x[..., row0:row0 + rows])  # without an input node the drive is per neuron
            for _ in range(int(rng.integers(2, 6))):
                op = ["run", "run_nograd", "reset", "write", "forward", "fit"][int(rng.integers(0, 6))]
                log.append(op)
                dev = 0.0
                drive = 2.0 if model == "qif" else 0.0
                if op in ("run", "run_nograd"):
                    T = int(rng.integers(30, 120)) if model == "qif" else int(rng.integers(2, 20))
                    x = (rng.standard_normal((T, width)) * 0.5 + drive).astype(np.float32)
                    kw = dict(sampling_steps=int(rng.integers(1, 4)), cutoff=int(rng.integers(0, 3)), verbose=False,
                              enable_grad=op == "run")
                    a, b = full.run(x, **kw).to_numpy("out"), sh.run(sl_in(x), **kw).to_numpy("out")
                    dev = float(np.abs(a - b).max()) if a.size and a.shape == b.shape else (0.0 if a.shape == b.shape else np.inf)
                elif op == "reset":
                    y = rng.uniform(-0.9, 0.9, full.state["rnn"].shape[0]).astype(np.float32)
                    full.reset({"rnn": y})
                    parts = y.reshape(-1, n)[:, row0:row0 + rows].reshape(-1)  # [v | s] -> this rank's [v | s]
                    sh.reset({"rnn": parts})
                elif op == "write":
                    if model == "qif":
                        continue  # (a spiking population exposes its lagging copy; nothing to write through)
                    v = (rng.standard_normal(n) * 0.5).astype(np.float32)
                    with torch.no_grad():
                        full.state["rnn"][:] = torch.tensor(v)
                        sh.state["rnn"][:] = torch.tensor(v[row0:row0 + rows])
                elif op == "forward":
                    full.compile()
                    sh.compile()
                    for _ in range(int(rng.integers(1, 4))):
                        x = (rng.standard_normal(width) * 0.5 + drive).astype(np.float32)
                        with torch.no_grad():
                            a = full.forward(torch.tensor(x)).numpy()
                            b = sh.forward(torch.tensor(sl_in(x))).numpy()
                        dev = max(dev, float(np.abs(a - b).max()))
                else:
                    w = int(rng.integers(3, 9))
                    T = w * int(rng.integers(1, 3))
                    x = (rng.standard_normal((T, width)) * 0.5 + drive).astype(np.float32)
                    tgt = (rng.standard_normal((T, k)) * 0.2).astype(np.float32)
                    kw = dict(optimizer=["sgd", "adam"][int(rng.integers(0, 2))], lr=1e-2, update_steps=w, sampling_steps=1,
                              verbose=False)
                    with quiet:
                        a, b = full.fit_bptt(x, tgt, **kw).to_numpy("out"), sh.fit_bptt(sl_in(x), tgt, **kw).to_numpy("out")
                    dev = float(np.abs(a - b).max())
                    trained = True
                    for name in train or []:
                        pa, pb = full.get_node("rnn")[name].detach().numpy(), sh.get_node("rnn")[name].detach().numpy()
                        pa = pa[row0:row0 + rows]
                        dev = max(dev, float(np.abs(pa - pb).max() / max(1.0, np.abs(pa).max())) / 10.0)  # (2e-4 relative)
                    if gd:
                        wa = full.get_edge("rnn", "out").weights.detach().numpy()[:, row0:row0 + rows]
                        dev = max(dev, float(np.abs(wa - sh.get_edge("rnn", "out").weights.detach().numpy()).max()) / 10.0)
                sa = full.state["rnn"].detach().numpy().reshape(-1, n)[:, row0:row0 + rows].reshape(-1)
                sb = sh.state["rnn"].detach().numpy()
                if trained or model == "qif":
                    if model == "qif":
                        sa, sb = sa[rows:], sb[rows:]  # compare s (v jumps by the reset when a spike lands a step apart)
                    dev = max(dev, float(np.abs(sa - sb).max()))
                elif not np.array_equal(sa, sb):
                    dev = np.inf
                worst = max(worst, dev)
                if not dev <= 2e-5:
                    print(f"rank {rank}: MISMATCH {dev} case {case} {log} n={n} m={m} k={k} {model} train={train} gd={gd}", flush=True)
                    fails[rank] = 1
                    return
        if rank == 0:
            print(f"{cases} random sharded call sequences on {world} ranks: worst deviation {worst:.3e}", flush=True)
    finally:
        dist.destroy_process_group()


def main(cases=20, seed=0, world=2):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    fails = mp.get_context("spawn").Array("i", [0] * world)
    mp.spawn(worker, args=(port, world, cases, seed, fails), nprocs=world, join=True)
    return int(any(fails))


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    sys.exit(main(*a))
