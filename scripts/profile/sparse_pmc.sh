#!/bin/bash
# What bounds the packed sparse step (r04 review, weak 6): separate rocprofv3 --pmc passes (with --kernel-trace only) of
# scripts/probes/sparse_pmc.py.  Summary -> gpurun_out/prof_sparse/summary.json (copy to profiles/).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_sparse
rm -rf $OUT && mkdir -p $OUT
P=/tmp/prof_sparse
rm -rf $P
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/insts -- python3 scripts/probes/sparse_pmc.py > $OUT/insts.log 2>&1 || { tail -5 $OUT/insts.log; exit 1; }
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $P/active -- python3 scripts/probes/sparse_pmc.py > $OUT/active.log 2>&1 || { tail -5 $OUT/active.log; exit 1; }
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_WAVES --kernel-trace --output-format csv -d $P/lds -- python3 scripts/probes/sparse_pmc.py > $OUT/lds.log 2>&1 || { tail -5 $OUT/lds.log; echo "(lds pass failed: counters not available)"; }
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for pas in ("insts", "active", "lds"):
    fs = glob.glob(f"/tmp/prof_sparse/{pas}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        k = "blocked" if "step_sparse_blocked_kernel" in n else "per_row" if "step_sparse_kernel" in n else None
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f"/tmp/prof_sparse/{pas}/**/*kernel_trace.csv", recursive=True)[0])):
        n = r["Kernel_Name"]
        k = "blocked" if "step_sparse_blocked_kernel" in n else "per_row" if "step_sparse_kernel" in n else None
        if k:
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, d in agg.items():
        c = {n: sorted(v)[len(v) // 2] for n, v in d.items()}
        c["median_duration_us_under_pmc"] = sorted(dur[k])[len(dur[k]) // 2] / 1e3
        out.setdefault(k, {})[pas] = c
json.dump(out, open("gpurun_out/prof_sparse/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
