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
    return hits[0] if hits else None


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

for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    per_kernel = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == counter:
            per_kernel[r.get("Kernel_Name", "")[:90]].append(float(r.get("Counter_Value", 0)))
    print(f"== {counter} per dispatch (raw counter units, KiB) ==")
    for name, vals in sorted(per_kernel.items(), key=lambda kv: -sum(kv[1]))[:6]:
        avg = sum(vals) / len(vals)
        print(f"{name:90s} n={len(vals):5d} avg={avg:14.1f}")
        summary.setdefault(name, {})[counter + "_avg_KiB"] = avg

json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
