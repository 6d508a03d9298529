"""HBM bytes per launch (FETCH_SIZE x 2 on gfx950 for 16-byte streaming reads + WRITE_SIZE, KiB units) against the
algorithmic bytes, per kernel, from the two PMC passes of scripts/archive/profile_kernels.sh."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
algo = {}
for line in open(os.path.join(out, "fetch.log")):
    if line.startswith("{"):
        algo = json.loads(line)["algorithmic_bytes_per_launch"]


def counters(sub, counter):
    per = defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                per[r.get("Kernel_Name", "")].append(float(r.get("Counter_Value", 0)))
    return per


fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
res = {}
for key, bytes_algo in algo.items():
    names = [n for n in fetch if key in n]
    if not names:
        continue
    name = max(names, key=lambda n: sum(fetch[n]))
    f = sorted(fetch[name])[len(fetch[name]) // 2]          # median launch
    w = sorted(write.get(name, [0.0]))[len(write.get(name, [0.0])) // 2]
    hbm = f * 1024 * 2 + w * 1024
    res[key] = dict(kernel=name[:110], launches=len(fetch[name]), FETCH_SIZE_raw_KiB=f, WRITE_SIZE_raw_KiB=w,
                    hbm_bytes_per_launch=hbm, algorithmic_bytes_per_launch=bytes_algo, ratio=hbm / bytes_algo)
    print(f"{key:22s} launches={len(fetch[name]):3d} hbm={hbm / 1e9:8.4f} GB  algorithmic={bytes_algo / 1e9:8.4f} GB  ratio={hbm / bytes_algo:.4f}")
res["correction"] = ("gfx950: FETCH_SIZE reports 1/2 of a 16 B/lane coalesced streaming read (MI355X_MICROARCH.md, HBM) -> "
                     "doubled; WRITE_SIZE exact; separate --pmc passes")
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
