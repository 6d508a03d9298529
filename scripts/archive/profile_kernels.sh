#!/bin/bash
# Per-kernel HBM traffic (rocprofv3 PMC, separate FETCH_SIZE / WRITE_SIZE passes as the MI355X guide prescribes) for the
# step, adjoint, QIF and RLS kernels at HBM-bound sizes.  Run on the GPU box; summary -> gpurun_out/prof_kernels/.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_kernels
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 scripts/pmc_kernels.py > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 scripts/pmc_kernels.py > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
python3 scripts/summarize_kernels.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
