#!/bin/bash
set -e
mkdir -p gpurun_out/r04j
python -m pytest tests/test_batched.py tests/test_coupled.py tests/test_abi.py -x -q -m gpu > gpurun_out/r04j/tests.log 2>&1 || { tail -40 gpurun_out/r04j/tests.log; exit 1; }
tail -3 gpurun_out/r04j/tests.log
