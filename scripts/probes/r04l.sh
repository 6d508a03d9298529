#!/bin/bash
set -e
mkdir -p gpurun_out/r04l
python3 scripts/probes/b32_cache.py > gpurun_out/r04l/b32.log 2>&1 || { tail -20 gpurun_out/r04l/b32.log; exit 1; }
cat gpurun_out/r04l/b32.log
