set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_gemm
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc -- python3 scripts/bench_gemm.py > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, json
f = glob.glob("gpurun_out/prof_gemm/pmc/**/*counter_collection.csv", recursive=True)[0]
tr = glob.glob("gpurun_out/prof_gemm/pmc/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(tr)):
    if "outer_accum" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), int(r["Grid_Size_X"]))
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "outer_accum" in r["Kernel_Name"]:
        rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
for d, c in rows.items():
    ns, grid = dur[d]
    if grid < 60000 * 256:
        continue
    clk = c["GRBM_GUI_ACTIVE"] / 8 / ns
    print(json.dumps({"grid_wgs": grid // 256, "ms": ns / 1e6, "clock_GHz": round(clk, 3),
                      "mfma_util": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 3),
                      "wait_any": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3), "wait_inst": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                      "active": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)}))
PY
