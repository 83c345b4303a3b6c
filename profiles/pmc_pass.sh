#!/bin/bash
# Usage (on the GPU box, via gpurun):  bash profiles/pmc_pass.sh <tag> <counter> [<counter> ...]
# One rocprofv3 --pmc pass (counters in their own run, with --kernel-trace only) over a short bench.
# Prints the per-kernel mean of every counter (last 15 launches) and leaves the csv under gpurun_out/pmc_<tag>.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
rm -rf $R/gpurun_out/pmc_$tag
cd /tmp
timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- \
    python3 $R/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-kernel-timing --no-explicit-sweep ${BENCH_ARGS} > $R/gpurun_out/pmc_$tag.log 2>&1
# what the counters were taken ON: the kernel sources' hash and the device (bench.py flags stale figures by the hash)
python3 - $R $R/gpurun_out/pmc_$tag.meta.json <<'PY'
import json, subprocess, sys
sys.path.insert(0, sys.argv[1])
from antsrl_amd.build import source_hash
try:
    uid = [l.split(":")[-1].strip() for l in subprocess.run(["rocm-smi", "--showuniqueid"], capture_output=True, text=True).stdout.splitlines() if "GPU[0]" in l][0]
except Exception:
    uid = None
json.dump(dict(source_sha16=source_hash(), device=uid), open(sys.argv[2], "w"))
PY
python3 - $R/gpurun_out/pmc_$tag <<'PY'
import collections, csv, glob, sys
fs = sorted(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"))
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[-1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    if k.startswith("k_"):
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    t = v[-15:]
    print("%-16s %-40s %.6g" % (k, c, sum(t) / len(t)))
PY
