#!/bin/bash
set -e
mkdir -p gpurun_out/r04d
export TMPDIR=/tmp
python3 scripts/probes/feedback_leg.py > gpurun_out/r04d/leg.json 2> gpurun_out/r04d/leg.err
cat gpurun_out/r04d/leg.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04d/prof -- python3 scripts/probes/feedback_leg.py > gpurun_out/r04d/prof.log 2>&1
f=$(find gpurun_out/r04d/prof -name "*kernel_stats.csv" | head -1)
head -12 $f | cut -c1-200
