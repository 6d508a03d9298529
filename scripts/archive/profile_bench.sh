#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box via gpurun): kernel-trace stats, then HBM counters in
# separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
# the stats pass profiles the driver's own command line
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
find $OUT -name "*.csv" | head -30
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
