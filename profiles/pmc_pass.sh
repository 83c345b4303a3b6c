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
