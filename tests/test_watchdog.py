"""A dead or hanging peer must not leave the other ranks of a row-sharded run waiting for ever (r04 review): the direct
exchange's hipStreamWaitValue32 and the host loop's collectives have no deadline of their own, so ShardedPopulation runs a
host-side StallWatchdog.  CPU rehearsal with two gloo ranks: one hangs, the other must exit with the watchdog's code
within the deadline (plus polling slack)."""
import os
import subprocess
import sys
import time

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_a_hanging_peer_ends_the_waiting_rank_within_the_deadline(oracle_backend):
    from rectipy_amd.sharded import StallWatchdog
    deadline = 2.0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2", DEADLINE=str(deadline),
               OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_stall_rank.py")], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    try:
        t0 = time.monotonic()
        out, err = procs[0].communicate(timeout=120)
        waited = time.monotonic() - t0
        assert procs[0].returncode == StallWatchdog.EXIT_CODE, (procs[0].returncode, out, err[-2000:])
        assert "three steps done" in out and "did not fire" not in out
        assert "rank stalled: no progress of the host-loop exchange" in err
        assert procs[1].poll() is None, "rank 1 is still hanging: it was the watchdog, not a closed connection, that ended rank 0"
        assert waited < 60, waited   # (start-up + three steps + deadline + polling slack)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()


@pytest.mark.gpu
def test_a_hanging_peer_of_the_direct_exchange_ends_the_waiting_rank(hip_device):
    """The stream-ordered direct exchange on the GPU (two rank processes on the one GPU of the box): rank 1 stops stepping,
    rank 0's stream waits in hipStreamWaitValue32 for its sequence number -- the watchdog reads the flag array on a side
    stream, sees no progress while the compute stream is busy and ends the process."""
    from rectipy_amd.sharded import StallWatchdog
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2", DEADLINE="3", STALL_DEVICE="cuda",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_stall_rank.py")], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    try:
        out, err = procs[0].communicate(timeout=180)
        assert procs[0].returncode == StallWatchdog.EXIT_CODE, (procs[0].returncode, out, err[-2000:])
        assert "three steps done" in out and "did not fire" not in out
        assert "no progress of the direct exchange" in err
        assert procs[1].poll() is None
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()


def test_watchdog_is_quiet_while_the_rank_is_idle_or_progressing():
    """No false alarms: idle (not busy) for longer than the deadline, and busy with progress, both leave the process alive."""
    from rectipy_amd.sharded import StallWatchdog
    state = dict(busy=False, n=0)
    dog = StallWatchdog(0.3, lambda: state["n"], lambda: state["busy"], "test")
    try:
        time.sleep(0.8)            # idle
        state["busy"] = True
        for _ in range(8):         # busy, progressing
            state["n"] += 1
            time.sleep(0.1)
        state["busy"] = False
        time.sleep(0.5)
    finally:
        dog.stop()
