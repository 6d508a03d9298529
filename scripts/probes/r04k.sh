#!/bin/bash
set -e
mkdir -p gpurun_out/r04k
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04k/bench.json 2> gpurun_out/r04k/bench.err || { tail -30 gpurun_out/r04k/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04k/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"])
e = d["extra"]
print("first_call", e["first_call"])
print("batched", {k: (round(v["ms_per_step"], 4), round(v.get("roof", {}).get("frac", v.get("roof_frac", 0)), 3)) for k, v in e["batched_trajectories"].items() if isinstance(v, dict)})
print("run_batch", e["batched_trajectories"]["Network_run_batch_B32"])
print("stepwise", e["stepwise"]["sizes"])
print("small_n", e["small_n"])
q = dict(e["qif_n65536"]); q.pop("workload"); print("qif", q)
print("rls", e["rls_n32768"])
PY
