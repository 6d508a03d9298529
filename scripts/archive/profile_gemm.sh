#!/bin/bash
# MFMA utilisation / LDS conflicts of the dW GEMM (separate --pmc pass; run on the GPU box via gpurun).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_gemm
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_gemm/pmc/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    key = "outer_accum" if "outer_accum" in name else "step_kernel" if "step_kernel" in name else "wt_partial" if "wt_partial" in name else None
    if key:
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: (max(v) if k == "outer_accum" else sum(v) / len(v)) for c, v in d.items()})
g = agg["outer_accum"]
if g:
    busy, total = max(g["SQ_VALU_MFMA_BUSY_CYCLES"]), max(g["SQ_BUSY_CYCLES"])
    print("outer_accum (T=1000 window): MFMA busy / SQ busy cycles =", busy / total if total else None,
          " LDS bank conflict / LDS active =", max(g["SQ_LDS_BANK_CONFLICT"]) / max(1.0, max(g["SQ_LDS_IDX_ACTIVE"])))
PY
