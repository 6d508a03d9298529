#!/bin/bash
set -e
mkdir -p gpurun_out/r04m
python -m pytest tests/test_rls.py tests/test_abi.py tests/test_edges.py -x -q -m gpu > gpurun_out/r04m/tests.log 2>&1 || { tail -40 gpurun_out/r04m/tests.log; exit 1; }
tail -3 gpurun_out/r04m/tests.log
python - <<'PY'
import json, torch, bench
print(json.dumps(bench.rls_extra(torch.device("cuda:0")), indent=1)[:1500])
PY
