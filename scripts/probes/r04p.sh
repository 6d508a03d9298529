#!/bin/bash
set -e
mkdir -p gpurun_out/r04p
python -m pytest tests/test_coupled.py tests/test_topologies.py tests/test_hip_parity.py -x -q -m gpu > gpurun_out/r04p/tests.log 2>&1 || { tail -60 gpurun_out/r04p/tests.log; exit 1; }
tail -3 gpurun_out/r04p/tests.log
python3 scripts/probes/feedback_leg.py 2> gpurun_out/r04p/leg.err | python3 -c "import json,sys; d=json.load(sys.stdin); print({k:v['events_us_per_step'] for k,v in d.items()})"
RECTIPY_AMD_COUPLED_LAUNCHES=member python3 scripts/probes/feedback_leg.py 2> gpurun_out/r04p/leg2.err | python3 -c "import json,sys; d=json.load(sys.stdin); print('per member', {k:v['events_us_per_step'] for k,v in d.items()})"
