#!/bin/bash
# MFMA utilisation, clock, wait cycles and LDS conflicts of the batched matrix-core steps (separate --pmc passes with
# --kernel-trace only; GPU box via gpurun).  Summary -> gpurun_out/prof_batched/summary.json
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_batched
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -- python3 scripts/archive/pmc_batched.py "$@" > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/wait -- python3 scripts/archive/pmc_batched.py "$@" > $OUT/wait.log 2>&1 || { tail -5 $OUT/wait.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, json, re
out = {}
for pas in ("pmc", "wait"):
    f = glob.glob(f"gpurun_out/prof_batched/{pas}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    def key(name):
        m = re.search(r"step_batched_mfma16_kernel<([^>]*)>", name)
        if m:
            return "mfma16<" + m.group(1).replace(" ", "") + ">"
        m = re.search(r"step_batched_mfma16_split_kernel<([^>]*)>", name)
        if m:
            return "mfma16_split<" + m.group(1).replace(" ", "") + ">"
        if "step_batched_mfma_kernel" in name:
            return "mfma32"
        return None
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    tr = glob.glob(f"gpurun_out/prof_batched/{pas}/**/*kernel_trace.csv", recursive=True)[0]
    for r in csv.DictReader(open(tr)):
        k = key(r["Kernel_Name"])
        if k:
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, d in agg.items():
        c = {n: sorted(v)[len(v) // 2] for n, v in d.items()}
        ns = sorted(dur[k])[len(dur[k]) // 2]
        c["median_duration_us_under_pmc"] = ns / 1e3
        c["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / ns
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            c["mfma_pipe_utilisation"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
            c["lds_conflict_per_active"] = c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"])
            c["W_stream_TBps"] = 4.0 * 32768 * 32768 / ns / 1e3
        if "SQ_WAIT_ANY" in c:  # quad-cycles, disjoint: waiting (s_waitcnt / barrier), issue-stalled, issuing
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                c[n + "_frac_of_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
        out.setdefault(k, {})[pas] = c
for k, v in out.items():
    print(k, json.dumps(v))
json.dump(out, open("gpurun_out/prof_batched/summary.json", "w"), indent=1)
PY
