set -o pipefail
mkdir -p gpurun_out/r03m
timeout -k 10 600 python -m pytest tests/test_rls.py tests/test_abi.py -x -q -m gpu > gpurun_out/r03m/rls_tests.log 2>&1 || { tail -40 gpurun_out/r03m/rls_tests.log; exit 1; }
tail -2 gpurun_out/r03m/rls_tests.log
timeout -k 10 500 python scripts/archive/probes/rls_sym_cap.py > gpurun_out/r03m/cap.jsonl 2> gpurun_out/r03m/cap.err || { tail -20 gpurun_out/r03m/cap.err; exit 1; }
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03m
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $GRAFT_REPO_ROOT/scripts/archive/probes/rls_sym_trace.py > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then grep -i "rls" "$f" | cut -c1-200; else echo "no stats csv"; fi
