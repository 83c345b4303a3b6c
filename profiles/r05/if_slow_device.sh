#!/bin/bash
# Lands on whatever device the pool hands out.  A "slow device" of round 4 (c3 at 0.231 whatever is drawn) shows up as a
# tuner run whose eight standard trials sit on ONE level (placement_trials.walk_steps > 0): the line says whether the walk
# found a fast pair; then the device's memory is mapped (region_map_probe + two_colour_probe).
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/slowdev
id=$(rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -n 1 | sed 's/.*: *//')
python3 $R/bench.py --steps 60 --no-cpu-baseline --no-explicit-sweep 2>/dev/null > $R/gpurun_out/slowdev/bench_$id.json
python3 - $R/gpurun_out/slowdev/bench_$id.json $id <<'PY' > $R/gpurun_out/slowdev/verdict_$id.txt
import json, sys
d = json.load(open(sys.argv[1])); pt = d["config"]["placement_trials_ms_per_step"]
print("device %s: c3 %.4f ms/step, k_perceive %.4f; tuner: walk_steps %d, both_levels_seen %s, chosen '%s'" % (
    sys.argv[2], d["ms_per_step"], d["roofline"]["kernel_ms"]["k_perceive"], pt["walk_steps"], pt["both_levels_seen"], pt["pairs"][pt["chosen"]]))
for l, t, k in zip(pt["pairs"], pt["ms_per_step"], pt["observation_kernel_ms"]):
    print("   %-46s step %.4f  k_perceive %.4f" % (l, t, k))
print("WALKED" if pt["walk_steps"] else "no walk")
PY
cat $R/gpurun_out/slowdev/verdict_$id.txt
if grep -q "^WALKED" $R/gpurun_out/slowdev/verdict_$id.txt; then
  echo "every standard trial on one level: mapping the device's memory"
  timeout -k 10 300 python3 $R/profiles/r05/region_map_probe.py 40 5 > $R/gpurun_out/slowdev/region_map_$id.txt 2>&1; cat $R/gpurun_out/slowdev/region_map_$id.txt
  timeout -k 10 300 python3 $R/profiles/r05/two_colour_probe.py > $R/gpurun_out/slowdev/two_colour_$id.txt 2>&1; tail -14 $R/gpurun_out/slowdev/two_colour_$id.txt
fi
