#!/bin/bash
set -e
mkdir -p gpurun_out/r04o
for ex in rccl direct; do
python bench.py --gpus 1 --force-dist --exchange $ex --steps 40 --warmup 5 --no-cpu-baseline --no-extra --config5-neurons 0 > gpurun_out/r04o/bench_$ex.json 2> gpurun_out/r04o/bench_$ex.err || { tail -30 gpurun_out/r04o/bench_$ex.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r04o/bench_$ex.json')); print('$ex', d['value'], d['config']['exchange'][:60], d.get('extra'))"
done
