"""The cross-rank sum of the sharded adjoint in a FIXED order (rank 0, 1, ..., P - 1): the host loop (gloo ranks on the
CPU) and the stream-ordered direct exchange (RP_COMM_DIRECT with an inbox per rank; 2 and 4 rank processes on the one GPU
of the box) reproduce, BIT FOR BIT, the one-process mirror of the sweep (sharded.simulate_sharded_backward on the oracle)
-- the same from run to run, whatever the arrival order.  tanh and QIF (surrogate gradients)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
CASES = [("li_tanh", 256, 18), ("qif", 128, 40)]


def mirror(oracle_backend, model, N, T, world):
    """The sweep of `world` row shards in one process on the oracle: forward windows rank by rank with the all-gather done
    by hand, then simulate_sharded_backward."""
    from _sharded_adjoint_common import backward_kwargs, shard_population, window_data
    from rectipy_amd.sharded import simulate_sharded_backward
    W, params, v0, x, g_out, kw = window_data(N, T, model)
    shards = [shard_population(model, N, world, r, "cpu", oracle_backend, W, params, v0, kw) for r in range(world)]
    pops = [p for p, _ in shards]
    rows = N // world
    saved = [dict(v=torch.zeros(T, rows), r=torch.zeros(T, rows), acc=torch.zeros(T, rows)) for _ in range(world)]
    full = torch.cat([p.source_local for p in pops])
    for p in pops:
        p.source.copy_(full)
    for t in range(T):
        for r, (p, sl) in enumerate(shards):
            p.forward(1, x=x[t:t + 1, sl].contiguous(), save_v=saved[r]["v"][t:t + 1], save_src=saved[r]["r"][t:t + 1],
                      save_acc=saved[r]["acc"][t:t + 1])
        full = torch.cat([p.source_local for p in pops])
        for p in pops:
            p.source.copy_(full)
    windows, results = [], []
    for r, (p, sl) in enumerate(shards):
        kw_b, res = backward_kwargs(model, p, T, "cpu", saved[r], g_out[:, sl].contiguous(), x[:, sl].contiguous())
        windows.append(kw_b)
        results.append(res)
    simulate_sharded_backward(pops, windows)
    return results


def _gloo_worker(rank, port, model, N, T, world, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from _sharded_adjoint_common import backward_kwargs, window_data
        from oracle.backend import OracleBackend
        from rectipy_amd.sharded import ShardedPopulation
        W, params, v0, x, g_out, kw = window_data(N, T, model)
        rows = N // world
        sl = slice(rank * rows, (rank + 1) * rows)
        if model == "qif":
            kw["s0_local"] = torch.zeros(rows)
        sp = ShardedPopulation(model, W[sl], {k: v[sl] for k, v in params.items()}, 1e-2, "cpu", n=N, v0_local=v0[sl],
                               backend=OracleBackend(), **kw)
        saved = dict(v=torch.zeros(T, rows), r=torch.zeros(T, rows), acc=torch.zeros(T, rows))
        for t in range(T):  # (save buffers per step: the host loop of ShardedPopulation.forward steps one at a time)
            sp.forward(1, x=x[t:t + 1, sl].contiguous(), save_v=saved["v"][t:t + 1], save_src=saved["r"][t:t + 1],
                       save_acc=saved["acc"][t:t + 1])
        kw_b, res = backward_kwargs(model, sp.pop, T, "cpu", saved, g_out[:, sl].contiguous(), x[:, sl].contiguous())
        args = dict(kw_b)
        sp.backward(args.pop("traj_v"), args.pop("traj_r"), args.pop("grad_out"), args.pop("grad_var"), args.pop("lam"),
                    args.pop("u_traj"), **args)
        np.savez(os.path.join(out_dir, f"{model}_{rank}.npz"), **{k: v.numpy() for k, v in res.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model,N,T", CASES)
@pytest.mark.parametrize("world", [2, 4])
def test_host_loop_adds_the_partials_in_rank_order(tmp_path, oracle_backend, model, N, T, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_gloo_worker, args=(port, model, N, T, world, str(tmp_path)), nprocs=world, join=True)
    ref = mirror(oracle_backend, model, N, T, world)
    for r in range(world):
        got = np.load(tmp_path / f"{model}_{r}.npz")
        for name, val in ref[r].items():
            assert np.array_equal(got[name], val.numpy()), (model, world, r, name)
        assert np.abs(got["lam"]).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_direct_exchange_adjoint_on_one_gpu_matches_the_mirror(tmp_path, oracle_backend, world):
    """2 / 4 rank processes on cuda:0: forward window through the direct all-gather, adjoint sweep through the direct
    rank-ordered sum (one library call each), against the one-process oracle mirror: bit for bit.  Run twice in the rank
    processes: the second sweep (other parities of the inbox, sequence numbers carried over) gives the same bits."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                   OUT_DIR=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_sharded_adjoint_rank.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank did not finish: the stream waits of the direct exchange were never satisfied")
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"SHARDED_ADJOINT_RANK_{rank}_OK" in out, out[-3000:]
    for model, N, T in CASES:
        ref = mirror(oracle_backend, model, N, T, world)
        for r in range(world):
            for rep in (0, 1):
                got = np.load(tmp_path / f"{model}_{r}_{rep}.npz")
                for name, val in ref[r].items():
                    assert np.array_equal(got[name], val.numpy()), (model, world, r, rep, name)
