#!/usr/bin/env python3
"""Condenses the rocprofv3 output of profiles/final_passes.sh (under gpurun_out/) into the tracked files:

    python profiles/summarize.py r03

  profiles/<round>/kernel_stats_<config>.csv   rocprofv3 --kernel-trace --stats, the step's kernels
  profiles/<round>/pmc_summary.md / .json       mean of the last 15 launches per kernel and counter
  profiles/traffic_<config>.json                HBM bytes per launch per kernel (read by bench.py)
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950
FETCH_SIZE tallies each 128-byte fabric read as 64 bytes (MI355X_MICROARCH.md, HBM section; confirmed in round 1
on k_sweep0: 520 MiB read, 260 MiB reported)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
out = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out, exist_ok=True)
summary = {}
for cfg in ("c3", "c2", "c4", "c5"):
    st = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_%s" % cfg, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if st:
        rows = list(csv.reader(open(st[-1])))
        keep = [rows[0]] + [r for r in rows[1:] if r[0].lstrip("void ").startswith("k_")]
        csv.writer(open(os.path.join(out, "kernel_stats_%s.csv" % cfg), "w")).writerows(keep)
    # the same trace restricted to the TIMED regions of the profiled bench run (its last steps x repeats launches of every
    # step kernel): rocprofv3's own stats average over the whole process, i.e. also over the 400 ageing steps whose first
    # ~150 are the faster start-of-episode transient — this file is the one to compare with the bench line's kernel_ms
    tr = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_%s" % cfg, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
    bl = os.path.join(ROOT, "gpurun_out", "bench_%s_rocprof.json" % cfg)
    if tr and os.path.exists(bl):
        try:
            line = [l for l in open(bl) if l.startswith("{")][-1]
            rec = json.loads(line)
            n_timed = int(rec["steps"]) * int(rec.get("repeats", 1))
            durs = collections.defaultdict(list)
            for r in csv.DictReader(open(tr[-1])):
                k = r["Kernel_Name"].replace("void ", "")
                if k.startswith("k_"):
                    durs[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            with open(os.path.join(out, "kernel_stats_%s_timed.csv" % cfg), "w") as f:
                w = csv.writer(f)
                w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs", "note"])
                for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
                    if len(v) >= n_timed:
                        t = v[-n_timed:]
                        w.writerow([k, len(t), "%.1f" % (sum(t) / len(t)), min(t), max(t),
                                    "last %d launches = the timed regions of bench.py --config %s (episode age %s)" %
                                    (n_timed, cfg, rec["config"].get("episode_age_steps"))])
            json.dump(rec, open(os.path.join(out, "bench_%s_under_rocprof.json" % cfg), "w"))
        except Exception as e:  # noqa: BLE001
            print("timed stats of %s: %s" % (cfg, e))
    per = {}
    for cname in ("fetch", "write"):
        fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (cfg, cname), "*", "*_counter_collection.csv")), key=os.path.getmtime)
        if not fs:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[-1])):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if k.startswith("k_") and not k.startswith(("k_reset", "k_gen", "k_collect", "k_copy", "k_read", "k_policy_w")):
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            if len(v) < 5:  # a kernel launched once or twice per run (the first step's k_move, the first update's k_update_one)
                continue
            t = v[-15:]
            per.setdefault(k, {})[c] = sum(t) / len(t)
    if per:
        summary[cfg] = per
        traffic = {k: int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024) for k, d in per.items() if "FETCH_SIZE" in d and "WRITE_SIZE" in d}
        metas = []
        for cname in ("fetch", "write"):
            mp = os.path.join(ROOT, "gpurun_out", "pmc_%s_%s.meta.json" % (cfg, cname))
            if os.path.exists(mp):
                metas.append(json.load(open(mp)))
        if metas and all(m == metas[0] for m in metas):  # (both passes on one tree and one device)
            traffic["_source_sha16"] = metas[0]["source_sha16"]
            traffic["_device"] = metas[0]["device"]
        traffic["_source"] = "profiles/%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, %s; steady regime: bench.py ages the episode 400 steps before the warm-up)" % (rnd, cfg)
        json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic_%s.json" % cfg), "w"), indent=1, sort_keys=True)
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
with open(os.path.join(out, "pmc_summary.md"), "w") as f:
    f.write("# %s PMC summary (rocprofv3 --pmc, one pass per counter, bench.py default age 400, --warmup 5 --steps 20)\n\n" % rnd)
    f.write("| config | kernel | FETCH_SIZE KiB (x2 = bytes read) | WRITE_SIZE KiB | HBM MB per launch |\n|---|---|---|---|---|\n")
    for cfg in sorted(summary):
        for k in sorted(summary[cfg]):
            d = summary[cfg][k]
            if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                f.write("| %s | %s | %.0f | %.0f | %.1f |\n" % (cfg, k, d["FETCH_SIZE"], d["WRITE_SIZE"], (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / 1e6))
print(json.dumps({c: {k: round((2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024 / 1e6, 1) for k, d in v.items()} for c, v in summary.items()}))
