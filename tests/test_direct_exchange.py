"""The stream-ordered direct exchange (RP_COMM_DIRECT) with TWO and with FOUR rank processes on the
one GPU of the box: each integrates its row shard, pushes its slice of the source vector into the other's buffer through a hipIpc
mapping, and the streams hand over with hipStreamWriteValue32 / hipStreamWaitValue32.  The assembled result must equal
the single-process run bit for bit (tanh and QIF; two library calls, so the sequence numbers carry over)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu_match_the_single_process_run(tmp_path, world):
    """world = 4: every rank pushes to three peers and waits on three flag slots per step -- the many-peer form of the
    protocol that an 8-GPU node runs with seven (the box admits at most 6 GPU processes: 4 ranks + this one)."""
    from _direct_exchange_rank import data
    from rectipy_amd import _abi
    from rectipy_amd.engine import Population
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                   OUT_DIR=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_direct_exchange_rank.py")], env=env,
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
        assert p.returncode == 0 and f"DIRECT_EXCHANGE_RANK_{rank}_OK" in out, out[-3000:]
    dev = torch.device("cuda:0")
    for model, N, T in (("li_tanh", 1024, 40), ("qif", 512, 60), ("li_tanh", 4096, 25)):
        W, params, v0, x, kw = data(N, T, model)
        if model == "qif":
            kw["s0"] = torch.zeros(N)
        full = Population(model, W, params, 1e-2, dev, v0=v0, **kw)
        traj = torch.zeros(T, N, device=dev)
        full.forward(T, x=x.to(dev), traj=traj, out_var=_abi.RP_OUT_SOURCE)
        parts = [np.load(tmp_path / f"{model}_{N}_{r}.npz") for r in range(world)]
        assert np.array_equal(np.concatenate([q["traj"] for q in parts], axis=1), traj.cpu().numpy()), model
        assert np.array_equal(np.concatenate([q["v"] for q in parts]), full.v.cpu().numpy()), model
        for q in parts:  # every rank ends with the complete, identical source vector
            assert np.array_equal(q["src"], full.source.cpu().numpy()), model
        if model == "qif":
            assert np.abs(np.diff(traj.cpu().numpy(), axis=0)).max() > 0.5, "no spikes in the QIF case"


def test_comm_struct_carries_the_direct_mode_fields():
    import ctypes as C

    from rectipy_amd import _abi, _lib
    lib = _lib.load()
    assert lib.rp_sizeof_comm() == C.sizeof(_abi.RpComm)
    assert [f[0] for f in _abi.RpComm._fields_][-7:] == ["mode", "seq", "peer_src", "peer_flags", "flags", "peer_inbox", "inbox"]
    assert _abi.RP_COMM_RCCL == 0 and _abi.RP_COMM_DIRECT == 1
    # argument checks that need no GPU
    assert lib.rp_ipc_alloc(0, None, None) == _abi.RP_ERR_ARG
    assert lib.rp_ipc_open(None, None) == _abi.RP_ERR_ARG
