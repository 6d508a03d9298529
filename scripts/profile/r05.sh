#!/bin/bash
# r05 profiling evidence (run on the GPU box via gpurun; --pmc passes separate, with --kernel-trace only).  Summaries land
# under gpurun_out/prof_r05/ (raw traces stay in /tmp on the box); copy what is to be judged into profiles/.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_r05
R=/tmp/prof_r05
rm -rf $OUT $R && mkdir -p $OUT $R
step() { echo "== $1 $(date +%T)"; }
stats_of() { f=$(find $1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $2; }

step "1 kernel stats of the driver's bench command (the child-process leg is skipped under the profiler)"
RECTIPY_AMD_BENCH_NO_CHILD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log || { tail -5 $OUT/stats.log; exit 1; }
step "2 HBM traffic of the step kernel (FETCH / WRITE, separate passes)"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/fetch -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/write -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
mkdir -p $R/sum && ln -s $R/stats $R/sum/stats && ln -s $R/fetch $R/sum/fetch && ln -s $R/write $R/sum/write
python3 scripts/summarize_prof.py $R > $OUT/summary.txt 2>&1; tail -30 $OUT/summary.txt
stats_of $R/stats $OUT/bench_kernel_stats.csv
cp $R/*.json $OUT/ 2>/dev/null
step "3 HBM traffic per kernel (scripts/pmc_kernels.py)"
K=$R/kernels; mkdir -p $K
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $K/fetch -- python3 scripts/pmc_kernels.py > $K/fetch.log 2>&1 || { tail -5 $K/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $K/write -- python3 scripts/pmc_kernels.py > $K/write.log 2>&1 || { tail -5 $K/write.log; exit 1; }
python3 scripts/summarize_kernels.py $K > $OUT/kernel_hbm_traffic.txt 2>&1; cat $OUT/kernel_hbm_traffic.txt; cp $K/summary.json $OUT/kernel_hbm_traffic.json
step "4 the coupled legs: run, fit, the LinearFilter edge, RLS"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/feedback -- python3 scripts/probes/feedback_leg.py > $OUT/feedback_leg.json 2> $OUT/feedback.log || tail -3 $OUT/feedback.log
stats_of $R/feedback $OUT/feedback_leg_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/fit -- python3 scripts/probes/feedback_fit_trace.py > $OUT/feedback_fit.json 2> $OUT/fit.log || tail -3 $OUT/fit.log
stats_of $R/fit $OUT/feedback_fit_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/filter -- python3 scripts/probes/filter_edge_leg.py > $OUT/filter_edge_leg.json 2> $OUT/filter.log || tail -3 $OUT/filter.log
stats_of $R/filter $OUT/filter_edge_leg_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/rls -- python3 scripts/probes/rls_leg.py > $OUT/rls_leg.json 2> $OUT/rls.log || tail -3 $OUT/rls.log
stats_of $R/rls $OUT/rls_blocked_kernel_stats.csv
step "5 counters of the packed sparse step"
bash scripts/profile/sparse_pmc.sh > $OUT/sparse_pmc.log 2>&1; cp gpurun_out/prof_sparse/summary.json $OUT/sparse_pmc.json
step "6 the plain driver command (no profiler) for the committed bench line"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_driver_cmd.json 2> $OUT/bench_line.err; cut -c1-600 $OUT/bench_line_driver_cmd.json
# trim the big per-kernel csv files to their library rows
for f in $OUT/*kernel_stats.csv; do python3 - "$f" <<'PY'
import csv, sys
f = sys.argv[1]
rows = list(csv.reader(open(f)))
keep = [rows[0]] + [r for r in rows[1:] if float(r[2]) > 2e5 or "anonymous" in r[0] or "rocblas" in r[0].lower() or "Cijk" in r[0]]
w = csv.writer(open(f, "w", newline=""))
for r in keep:
    r[0] = r[0][:160]
    w.writerow(r)
PY
done
step "done"
