#!/bin/bash
# Lands on whatever device the pool hands out; when it is one of the SLOW ones (c3 step > 0.22 ms with the tuner's best pair),
# maps its memory: is there ANY zone on it in which the observation tensor is fast?  (region_map_probe + two_colour_probe)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/slowdev
id=$(rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -n 1 | sed 's/.*: *//')
python3 $R/bench.py --steps 60 --no-cpu-baseline --no-explicit-sweep 2>/dev/null > $R/gpurun_out/slowdev/bench_$id.json
ms=$(python3 -c "import json,sys; d=json.load(open('$R/gpurun_out/slowdev/bench_$id.json')); print('%.4f' % d['ms_per_step']); print(d['config']['placement_trials_ms_per_step'], file=sys.stderr)")
echo "device $id: c3 $ms ms/step"
if python3 -c "import sys; sys.exit(0 if float('$ms') > 0.22 else 1)"; then
  echo "SLOW device: mapping its memory"
  timeout -k 10 300 python3 $R/profiles/r05/region_map_probe.py 40 5 > $R/gpurun_out/slowdev/region_map_$id.txt 2>&1; cat $R/gpurun_out/slowdev/region_map_$id.txt
  timeout -k 10 300 python3 $R/profiles/r05/two_colour_probe.py > $R/gpurun_out/slowdev/two_colour_$id.txt 2>&1; tail -12 $R/gpurun_out/slowdev/two_colour_$id.txt
fi
