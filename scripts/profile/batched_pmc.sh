#!/bin/bash
# Which unit bounds the 32-trajectory matrix-core step (r04 review, item 3): separate rocprofv3 --pmc passes (with --kernel-trace
# only; the program directly after --) over scripts/archive/pmc_batched.py on the measurement build: MFMA pipe busy and LDS
# conflicts, wait / issue cycles, and the L2 / HBM side (hits, misses, fetched bytes).  Summary -> gpurun_out/prof_batched/summary.json
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_batched
rm -rf $OUT && mkdir -p $OUT
P=/tmp/prof_batched
rm -rf $P
timeout -k 10 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/pmc -- python3 scripts/archive/pmc_batched.py > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
timeout -k 10 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/wait -- python3 scripts/archive/pmc_batched.py > $OUT/wait.log 2>&1 || { tail -5 $OUT/wait.log; exit 1; }
timeout -k 10 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $P/l2 -- python3 scripts/archive/pmc_batched.py > $OUT/l2.log 2>&1 || { tail -5 $OUT/l2.log; echo "(l2 pass failed: counters not available)"; }
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $P/issue -- python3 scripts/archive/pmc_batched.py > $OUT/issue.log 2>&1 || { tail -5 $OUT/issue.log; echo "(issue pass failed: counters not available)"; }
python3 - <<'PY'
import csv, glob, collections, json, re
out = {}
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
for pas in ("pmc", "wait", "l2", "issue"):
    fs = glob.glob(f"/tmp/prof_batched/{pas}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = key(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f"/tmp/prof_batched/{pas}/**/*kernel_trace.csv", recursive=True)[0])):
        k = key(r["Kernel_Name"])
        if k:
            dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, d in agg.items():
        c = {n: sorted(v)[len(v) // 2] for n, v in d.items()}
        ns = sorted(dur[k])[len(dur[k]) // 2]
        c["median_duration_us_under_pmc"] = ns / 1e3
        if "GRBM_GUI_ACTIVE" in c:
            c["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / ns
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            c["mfma_pipe_utilisation"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 256 * 4)
            c["lds_conflict_per_active"] = c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"])
            c["W_stream_TBps"] = 4.0 * 32768 * 32768 / ns / 1e3
        if "SQ_WAIT_ANY" in c:  # quad-cycles, disjoint: waiting (s_waitcnt / barrier), issue-stalled, issuing
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                c[n + "_frac_of_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
        if "TCC_HIT_sum" in c:
            c["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        out.setdefault(k, {})[pas] = c
for k, v in out.items():
    print(k, json.dumps(v))
json.dump(out, open("gpurun_out/prof_batched/summary.json", "w"), indent=1)
PY
