#!/usr/bin/env python3
"""Condenses rocprofv3 output under gpurun_out/ into the tracked files of profiles/<round>/.

    python profiles/summarize.py r01 prof_final f_fetch f_write f_sq [x_fetch x_write]

Writes  profiles/<round>/kernel_stats.csv        (rocprofv3 --kernel-trace --stats, hot kernels)
        profiles/<round>/pmc_summary.{md,json}   (mean of the last 15 launches per kernel/counter)
        profiles/traffic_c3.json                 (HBM bytes per launch per kernel, read by bench.py)
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB, and
on gfx950 FETCH_SIZE tallies each 128-byte fabric read as 64 bytes (MI355X_MICROARCH.md, HBM section;
confirmed here: k_sweep0 reads 520 MiB and reports 260 MiB, and TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ
for k_act's gathers, so the factor holds for the gather pattern too)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, stats_dir, *pmc = sys.argv[1:]
out = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out, exist_ok=True)
st = glob.glob(os.path.join(ROOT, "gpurun_out", stats_dir, "*", "*_kernel_stats.csv"))
if st:
    st.sort(key=os.path.getmtime)
    rows = [r for r in csv.reader(open(st[-1]))]
    keep = [rows[0]] + [r for r in rows[1:] if "k_" in r[0][:12] or r[0].startswith("void k_")]
    csv.writer(open(os.path.join(out, "kernel_stats.csv"), "w")).writerows(keep)
summary = {}
for tag in pmc:
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_" + tag, "*", "*_counter_collection.csv"))
    if not fs:
        continue
    agg = collections.defaultdict(list)
    fs.sort(key=os.path.getmtime)  # gpurun_out/ accumulates earlier runs: take the newest
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith(("k_act", "k_sweep", "k_update")):
            agg[(k.split("<")[0] + ("" if not tag.startswith("x_") else "@explicit"), r["Counter_Name"])].append(
                float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        tail = v[-15:]
        summary.setdefault(k, {})[c] = sum(tail) / len(tail)
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
with open(os.path.join(out, "pmc_summary.md"), "w") as f:
    f.write("# %s PMC summary (rocprofv3 --pmc, one pass per counter group, bench.py c3)\n\n" % rnd)
    f.write("| kernel | counter | mean of last 15 launches |\n|---|---|---|\n")
    for k in sorted(summary):
        for c in sorted(summary[k]):
            f.write("| %s | %s | %.6g |\n" % (k, c, summary[k][c]))
traffic = {}
for k, d in summary.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d and "@" not in k:
        traffic[k] = int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
    if "@explicit" in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d and k.startswith("k_sweep0"):
        traffic["k_sweep0"] = int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
if traffic:
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic_c3.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(traffic))
