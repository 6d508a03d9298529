set -o pipefail
mkdir -p gpurun_out/r03k
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03k
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $GRAFT_REPO_ROOT/scripts/archive/probes/rls_sym_trace.py > $OUT/trace.log 2>&1 || { tail -20 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then grep -i "rls" "$f" | cut -c1-220; else echo "no stats csv"; ls -R $OUT/trace | head; fi
