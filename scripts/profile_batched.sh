#!/bin/bash
# MFMA utilisation, clock and LDS conflicts of the batched matrix-core step (separate --pmc pass; GPU box via gpurun).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_batched
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -- python3 scripts/bench_batched.py 32768 > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, json
f = glob.glob("gpurun_out/prof_batched/pmc/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    key = "mfma" if "step_batched_mfma" in name else "valu_b8" if "step_batched_kernel<0, 4, 8" in name else "step_kernel" if "step_kernel<0" in name else None
    if key:
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
tr = glob.glob("gpurun_out/prof_batched/pmc/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(tr)):
    name = r["Kernel_Name"]
    key = "mfma" if "step_batched_mfma" in name else "valu_b8" if "step_batched_kernel<0, 4, 8" in name else "step_kernel" if "step_kernel<0" in name else None
    if key:
        dur[key].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for k, d in agg.items():
    c = {n: sorted(v)[len(v) // 2] for n, v in d.items()}
    ns = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else None
    c["median_duration_us_under_pmc"] = None if ns is None else ns / 1e3
    if ns:
        c["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / ns
    c["mfma_pipe_utilisation"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 256 * 4) if c.get("GRBM_GUI_ACTIVE") else None
    c["lds_conflict_per_active"] = c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"])
    out[k] = c
    print(k, json.dumps(c))
json.dump(out, open("gpurun_out/prof_batched/summary.json", "w"), indent=1)
PY
