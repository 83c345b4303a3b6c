export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rm -rf $R/gpurun_out/prof_dist
ANTSRL_BENCH_FORCE_DIST=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_dist -- python3 $R/bench.py --steps 40 --warmup 5 --repeats 1 --age 100 --no-cpu-baseline --no-explicit-sweep --no-kernel-timing --no-tune-placement > /dev/null 2> $R/gpurun_out/prof_dist.err
python3 - $R/gpurun_out/prof_dist <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%-40s q=%s start %8.1f us  dur %7.1f us" % (r["Kernel_Name"][:40], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
