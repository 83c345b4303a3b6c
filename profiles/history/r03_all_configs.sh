#!/bin/bash
# every bench configuration on one box, final tree of the round (default flags: age 400, 5 x 200 steps)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { python3 bench.py "$@" --no-cpu-baseline --no-explicit-sweep 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-28s ms/step=%.4f value=%.4g regions=%s %s frac=%s' % ('$*', d['ms_per_step'], d['value'], d['ms_per_step_regions'], d['roofline']['kernel_ms'], d['roofline']['frac']))"; }
run --config c3
run --config c2
run --config c4 --steps 100
run --config c5
run --config c5 --no-obs
run --config c1
run --config c3 --diffuse 0.02
run --config c3
python3 bench.py > gpurun_out/r03_bench_default1.json 2> gpurun_out/r03_bench_default1.err; tail -c 2400 gpurun_out/r03_bench_default1.json
