#!/bin/bash
# r04 profiling evidence (run on the GPU box via gpurun): (1) kernel-trace stats of the DRIVER's bench command, (2) HBM
# traffic of the step kernel from separate --pmc passes of a reduced bench run, (3) per-kernel HBM traffic of every
# streaming kernel (scripts/pmc_kernels.py), (4) kernel-trace stats of the FeedbackNetwork leg alone (no rocBLAS on it).
# Summaries land under gpurun_out/prof_r04/; copy what is to be judged into profiles/.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_r04
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log || { tail -5 $OUT/stats.log; exit 1; }
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
echo "bench passes done"
K=$OUT/kernels
mkdir -p $K
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $K/fetch -- python3 scripts/pmc_kernels.py > $K/fetch.log 2>&1 || { tail -5 $K/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $K/write -- python3 scripts/pmc_kernels.py > $K/write.log 2>&1 || { tail -5 $K/write.log; exit 1; }
python3 scripts/summarize_kernels.py $K > $K/summary.txt 2>&1
cat $K/summary.txt
echo "kernel passes done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/feedback -- python3 scripts/probes/feedback_leg.py > $OUT/feedback_leg.json 2> $OUT/feedback.log || { tail -5 $OUT/feedback.log; exit 1; }
f=$(find $OUT/feedback -name "*kernel_stats.csv" | head -1)
cp $f $OUT/feedback_kernel_stats.csv
head -8 $OUT/feedback_kernel_stats.csv | cut -c1-160
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f $OUT/bench_kernel_stats.csv
# the plain driver command without the profiler, for the committed bench line
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_driver_cmd.json 2> $OUT/bench_line.err
# the raw traces are large (gpurun merges at most 64 MiB back): keep the summaries only
rm -rf $OUT/stats $OUT/fetch $OUT/write $K/fetch $K/write $OUT/feedback
