#!/bin/bash
set -e
mkdir -p gpurun_out/r04h
python -m pytest tests/test_coupled.py -x -q -m gpu > gpurun_out/r04h/tests.log 2>&1 || { tail -40 gpurun_out/r04h/tests.log; exit 1; }
tail -3 gpurun_out/r04h/tests.log
for i in 1 2 3; do python3 scripts/probes/feedback_leg.py 2> gpurun_out/r04h/leg.err | python3 -c "import json,sys; d=json.load(sys.stdin); print({k:v['events_us_per_step'] for k,v in d.items()})"; done
