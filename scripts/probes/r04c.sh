#!/bin/bash
set -e
mkdir -p gpurun_out/r04c
cd /tmp && export TMPDIR=/tmp
cat > /tmp/leg.py <<'PY'
import json, sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
print(json.dumps(bench.stepwise_extra(dev), indent=1))
PY
python3 /tmp/leg.py > /root/repo/gpurun_out/r04c/stepwise.json 2> /root/repo/gpurun_out/r04c/stepwise.err
cat /root/repo/gpurun_out/r04c/stepwise.json
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r04c/prof -o leg -- python3 /tmp/leg.py > /root/repo/gpurun_out/r04c/prof.log 2>&1
python3 /root/repo/scripts/summarize_prof.py /root/repo/gpurun_out/r04c/prof 2>/dev/null | head -40 || true
ls -R /root/repo/gpurun_out/r04c/prof | head -20
