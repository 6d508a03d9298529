#!/bin/bash
# r04 first GPU call: coupled-population tests + stepwise leg timing
set -e
mkdir -p gpurun_out/r04a
python -m pytest tests/test_coupled.py tests/test_topologies.py tests/test_abi.py -x -q -m gpu > gpurun_out/r04a/tests.log 2>&1 || { tail -40 gpurun_out/r04a/tests.log; exit 1; }
tail -3 gpurun_out/r04a/tests.log
python - <<'PY' > gpurun_out/r04a/stepwise.json 2> gpurun_out/r04a/stepwise.err
import json, torch, bench
dev = torch.device("cuda:0")
print(json.dumps(bench.stepwise_extra(dev), indent=1))
PY
cat gpurun_out/r04a/stepwise.json
