"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into the small summaries kept under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pattern):
    hits = glob.glob(os.path.join(out, sub, "**", pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None  # (gpurun merges into gpurun_out/: older runs' files stay)


stats = find("stats", "*kernel_stats.csv")
summary = {}
if stats:
    rows = list(csv.DictReader(open(stats)))
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    print(f"{'kernel':90s} {'calls':>7s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    for r in rows[:12]:
        name = r.get("Name", "")[:90]
        calls = int(r.get("Calls", 0))
        avg = float(r.get("AverageNs", 0)) / 1e3
        tot = float(r.get("TotalDurationNs", 0)) / 1e6
        print(f"{name:90s} {calls:7d} {avg:10.2f} {tot:10.2f} {float(r.get('Percentage', 0)):6.2f}")
        summary[name] = dict(calls=calls, avg_us=avg, total_ms=tot)

# the same kernel instance runs on several grids (other shapes of the bench's extra legs, and the few hundred launches
# of rp_tune_forward's measurement): per (kernel, grid) durations from the trace of the same run
trace = find("stats", "*kernel_trace.csv")
if trace:
    per = defaultdict(list)
    for r in csv.DictReader(open(trace)):
        if "step_kernel<" in r.get("Kernel_Name", ""):
            per[(r["Kernel_Name"][:60], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    # the headline instance also runs on other SHAPES with the same grid (the N/2 populations of extra.stepwise): keep the
    # launches of the benchmark shape apart by their duration (16384^2: ~150 us, 32768^2: ~600 us)
    head = [v for (name, grid), v in per.items() if "step_kernel<0, 2, 1, true, true>" in name and grid == 683]
    if head:
        big = [d for d in head[0] if d >= 400.0]
        if big:
            print(f"== step_kernel<0,2,1,vec4,nt> on 683 workgroups, launches of the N=32768 shape only (>= 400 us): "
                  f"calls={len(big)} avg_us={sum(big) / len(big):.2f} min_us={min(big):.2f} max_us={max(big):.2f} ==")
            summary["step_kernel<0,2,1,true,true> @ 683 workgroups, N=32768 launches (>= 400 us)"] = dict(
                calls=len(big), avg_us=sum(big) / len(big), min_us=min(big), max_us=max(big))
    print("== step kernel per (instance, workgroups) from the kernel trace ==")
    print(f"{'kernel':60s} {'workgroups':>10s} {'calls':>7s} {'avg_us':>10s} {'min_us':>10s}")
    for (name, grid), v in sorted(per.items(), key=lambda kv: -len(kv[1]))[:6]:
        print(f"{name:60s} {grid:10d} {len(v):7d} {sum(v) / len(v):10.2f} {min(v):10.2f}")
        summary[f"{name} @ {grid} workgroups"] = dict(calls=len(v), avg_us=sum(v) / len(v), min_us=min(v))

for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    per_kernel = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == counter:
            per_kernel[r.get("Kernel_Name", "")[:90]].append(float(r.get("Counter_Value", 0)))
    print(f"== {counter} per dispatch (raw counter units, KiB) ==")
    ranked = sorted(per_kernel.items(), key=lambda kv: -sum(kv[1]))
    for name, vals in ranked[:6] + [kv for kv in ranked[6:] if "step_kernel<" in kv[0]]:
        avg = sum(vals) / len(vals)
        print(f"{name:90s} n={len(vals):5d} avg={avg:14.1f}")
        summary.setdefault(name, {})[counter + "_avg_KiB"] = avg

json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
