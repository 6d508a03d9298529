set -o pipefail
mkdir -p gpurun_out/r03i
timeout -k 10 900 python -m pytest tests/test_rls.py -x -q -m gpu > gpurun_out/r03i/rls_tests.log 2>&1 || { tail -30 gpurun_out/r03i/rls_tests.log; exit 1; }
tail -3 gpurun_out/r03i/rls_tests.log
timeout -k 10 600 python scripts/archive/probes/rls_sym_rows.py > gpurun_out/r03i/rows.jsonl 2> gpurun_out/r03i/rows.err || { tail -20 gpurun_out/r03i/rows.err; exit 1; }
cat gpurun_out/r03i/rows.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03i/trace -o run -- python3 $GRAFT_REPO_ROOT/scripts/archive/probes/rls_sym_trace.py > $GRAFT_REPO_ROOT/gpurun_out/r03i/trace.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/r03i/trace.log; exit 1; }
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r03i/trace -name "*kernel_stats.csv" | head -1); grep -i "rls" $f | cut -c1-200
