"""`python bench.py --gpus N` without a launcher starts its own ranks (bench.self_launch): fresh children through
torch.distributed.run on a free port, rank 0's JSON line relayed as the only stdout line, the children's exit status
returned.  Runs on the CPU: the rank processes here are stub scripts, so no GPU is touched."""
import json
import os
import subprocess
import sys
import textwrap

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STUB_OK = textwrap.dedent("""
    import json, os, sys
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    print(f"rank {rank} of {world} says hello")          # noise on stdout: must not reach the parent's stdout
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": world, "argv": sys.argv[1:]}))
""")

STUB_FAIL = textwrap.dedent("""
    import json, os, sys
    if int(os.environ["RANK"]) == 1:
        sys.exit(7)
    print(json.dumps({"metric": "stub", "value": 1.0}))
""")


def _launch(tmp_path, capfd, body, argv):
    import bench
    stub = tmp_path / "stub_rank.py"
    stub.write_text(body)
    rc = bench.self_launch(2, argv, script=str(stub), timeout=300)
    cap = capfd.readouterr()
    return rc, cap.out, cap.err


def test_self_launch_relays_rank0_json_only(tmp_path, capfd):
    rc, out, err = _launch(tmp_path, capfd, STUB_OK, ["--gpus", "2", "--steps", "3"])
    assert rc == 0, err[-2000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out
    rec = json.loads(lines[0])
    assert rec["metric"] == "stub" and rec["n_gpus"] == 2 and rec["argv"] == ["--gpus", "2", "--steps", "3"]
    assert "says hello" in err and "launching" in err


def test_self_launch_propagates_a_failed_rank(tmp_path, capfd):
    rc, out, err = _launch(tmp_path, capfd, STUB_FAIL, ["--gpus", "2"])
    assert rc != 0
    assert out.strip() == "", "no JSON line may be relayed from a job in which a rank failed"


def test_bench_gpus_2_reaches_rank_startup():
    """The parent path of the real script: `python bench.py --gpus 2` with no WORLD_SIZE becomes the launcher instead of
    exiting on the argument check.  Without GPUs the ranks die at their first GPU call, and that status comes back."""
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip("would run the real 2-rank benchmark on this box")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                           "--neurons", "256", "--config5-neurons", "0"], capture_output=True, text=True, env=env, timeout=600)
    assert "launching" in proc.stderr and "torch.distributed.run" in proc.stderr
    assert "launch with torch.distributed.run" not in proc.stderr  # the old SystemExit of the argument check
    assert proc.returncode != 0 and proc.stdout.strip() == ""  # ranks started and failed at torch.cuda.set_device


def test_mismatched_world_size_is_still_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                          env=env, timeout=300)
    assert proc.returncode != 0 and "--nproc-per-node must equal --gpus" in proc.stderr
