#!/bin/bash
# PMC of the dense / padded observation layouts (profiles/r04/obs_stride_ab.py, ONE process: the two arms are two template
# instances of k_perceive, told apart by name).  Separate passes, --kernel-trace only.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for pass in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_PENDING_STALL_CYCLES_sum SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_stride_$tag
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_stride_$tag -- \
      python3 $R/profiles/r04/obs_stride_ab.py --blocks 1 --steps 20 > $R/gpurun_out/pmc_stride_$tag.log 2>&1
  python3 - $R/gpurun_out/pmc_stride_$tag <<'PY'
import collections, csv, glob, sys
fs = sorted(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"))
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[-1])):
    n = r["Kernel_Name"]
    if "k_perceive" not in n: continue
    arm = "padded" if n.split("<")[1].split(">")[0].replace(" ", "").endswith("true") else "dense"
    agg[(arm, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    t = v[-15:]
    print("k_perceive %-7s %-32s %.6g  (n=%d)" % (k, c, sum(t) / len(t), len(v)))
PY
done
