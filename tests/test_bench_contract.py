"""bench.py prints exactly one JSON line with the fields the driver reads (GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract():
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "20",
                           "--neurons", "8192", "--no-cpu-baseline", "--qif-steps", "400"], capture_output=True, text=True, env=env, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"expected one line on stdout, got {len(lines)}"
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["unit"] == "steps/s" and out["n_gpus"] == 1 and out["steps"] == 300 and out["warmup"] == 20
    assert out["vs_baseline"] is None and out["dtype"] == "f32" and out["data"] == "synthetic"
    assert "workload" in out["config"] and "model" not in out["config"]
    roof = out["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0.05 < roof["frac"] < 1.0
    assert abs(out["value"] - 1e3 / out["ms_per_step"]) / out["value"] < 1e-6
    assert out["state_finite"] is True
    extra = out["extra"]
    assert extra["grad_finite"] is True and extra["fwd_bwd_steps_per_s"] > 0
