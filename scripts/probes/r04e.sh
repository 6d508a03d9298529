#!/bin/bash
set -e
mkdir -p gpurun_out/r04e
python3 scripts/probes/coupled_placement.py > gpurun_out/r04e/placement.log 2>&1 || { tail -20 gpurun_out/r04e/placement.log; exit 1; }
cat gpurun_out/r04e/placement.log
