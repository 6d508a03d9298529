"""Reduce a rocprofv3 --kernel-trace CSV to a timeline of the library's kernels: per kernel family the launch durations in
launch order, printed as compact rows (us), plus the gaps between consecutive launches of the stream."""
import csv
import glob
import re
import sys

root = sys.argv[1]
files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |\(.*", "", n)[:48]
fam = {}
for s, e, n in rows:
    fam.setdefault(short(n), []).append((s, (e - s) / 1e3))
t0 = rows[0][0]
for k, v in sorted(fam.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    tot = sum(d for _, d in v)
    if tot < 500:
        continue
    print(f"== {k}: {len(v)} launches, total {tot / 1e3:.2f} ms, mean {tot / len(v):.1f} us")
    if len(v) >= 20 and ("step_" in k or "wt_partial" in k):
        for i in range(0, len(v), 20):
            print("   t=%8.1f ms: " % ((v[i][0] - t0) / 1e6) + " ".join(f"{d:5.0f}" for _, d in v[i:i + 20]))
