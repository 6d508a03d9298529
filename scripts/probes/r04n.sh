#!/bin/bash
set -e
mkdir -p gpurun_out/r04n
python -m pytest tests/test_sharded_adjoint.py tests/test_direct_exchange.py tests/test_rccl_inlib.py -x -q -m gpu > gpurun_out/r04n/tests.log 2>&1 || { tail -60 gpurun_out/r04n/tests.log; exit 1; }
tail -3 gpurun_out/r04n/tests.log
