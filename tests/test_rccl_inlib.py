"""In-library RCCL exchange (rp_forward_sharded / rp_backward_sharded) on the GPU box: one-rank RCCL group in a
child process (the default process group must not leak into the other tests).  Multi-rank arithmetic is covered by the
gloo tests in test_sharded.py; this one pins the RCCL call path itself (communicator hand-over, in-place all-gather,
reduce-scatter, stream ordering) against the unsharded library call."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_inlibrary_exchange_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_one_rank.py")], env=env, capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0 and "RCCL_INLIB_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


def test_comm_struct_matches_header():
    """rp_comm mirror: field order / size as declared in include/rectipy_hip.h (no GPU needed)."""
    import ctypes as C

    from rectipy_amd import _abi, _lib
    lib = _lib.load()
    assert lib.rp_sizeof_comm() == C.sizeof(_abi.RpComm)
    assert [f[0] for f in _abi.RpComm._fields_][:6] == ["comm", "all_gather", "reduce_scatter", "rank", "world", "work"]
