#!/bin/bash
set -e
mkdir -p gpurun_out/r04q
python -m pytest tests/test_coupled.py -x -q -m gpu > gpurun_out/r04q/tests.log 2>&1 || { tail -60 gpurun_out/r04q/tests.log; exit 1; }
tail -3 gpurun_out/r04q/tests.log
python3 scripts/probes/feedback_leg.py 2> gpurun_out/r04q/leg.err | python3 -c "import json,sys; d=json.load(sys.stdin); print({k:v['events_us_per_step'] for k,v in d.items()})"
RECTIPY_AMD_COUPLED_LAUNCHES=member python3 scripts/probes/feedback_leg.py 2> gpurun_out/r04q/leg2.err | python3 -c "import json,sys; d=json.load(sys.stdin); print('per member', {k:v['events_us_per_step'] for k,v in d.items()})"
