#!/bin/bash
set -e
mkdir -p gpurun_out/r04i
python scripts/bench_coupled.py > gpurun_out/r04i/coupled.log 2>&1 || { tail -30 gpurun_out/r04i/coupled.log; exit 1; }
grep -v "^{" gpurun_out/r04i/coupled.log
