#!/bin/bash
set -e
mkdir -p gpurun_out/r04g
python3 scripts/probes/coupled_alloc2.py > gpurun_out/r04g/alloc2.log 2>&1 || { tail -20 gpurun_out/r04g/alloc2.log; exit 1; }
cat gpurun_out/r04g/alloc2.log
