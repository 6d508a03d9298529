#!/bin/bash
set -e
mkdir -p gpurun_out/r04b
python scripts/bench_coupled.py > gpurun_out/r04b/coupled.log 2>&1 || { tail -30 gpurun_out/r04b/coupled.log; exit 1; }
cat gpurun_out/r04b/coupled.log
python -m pytest tests/test_abi.py tests/test_batched.py tests/test_rls.py tests/test_hip_parity.py -x -q -m gpu > gpurun_out/r04b/tests.log 2>&1 || { tail -40 gpurun_out/r04b/tests.log; exit 1; }
tail -3 gpurun_out/r04b/tests.log
