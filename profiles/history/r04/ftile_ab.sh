#!/bin/bash
# c4: 4 x 4-cell blocks for the {food, META} records of the explicit-sweep layout (KP::ftile) against row-major records
# (ANTSRL_NO_TILED=1, profiling library), alternating, separate processes.
R=${GRAFT_REPO_ROOT:-/root/repo}
export ANTSRL_LIB=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for i in 1 2 3; do
  for v in "X=1" "ANTSRL_NO_TILED=1"; do
    env $v python3 $R/bench.py --config ${CFG:-c4} --steps 60 --warmup 10 --age 200 --repeats 3 --no-cpu-baseline --no-explicit-sweep ${EXTRA} > /tmp/ftile.json 2>/dev/null
    python3 - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ftile.json").read().strip().splitlines()[-1])
print("%-20s %.4f ms/step  %s" % (sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
  done
done
