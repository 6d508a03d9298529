#!/bin/bash
set -e
mkdir -p gpurun_out/r04f
python3 scripts/probes/coupled_alloc.py > gpurun_out/r04f/alloc.log 2>&1 || { tail -20 gpurun_out/r04f/alloc.log; exit 1; }
cat gpurun_out/r04f/alloc.log
python3 scripts/probes/feedback_leg.py > gpurun_out/r04f/leg.json 2> gpurun_out/r04f/leg.err
cat gpurun_out/r04f/leg.json
