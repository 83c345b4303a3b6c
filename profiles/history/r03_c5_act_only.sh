#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/r03_tests1.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03_tests1.log
run() { python3 bench.py "$@" --no-cpu-baseline --no-explicit-sweep 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('$*', 'ms/step=%.4f value=%.4g %s' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))"; }
for i in 1 2; do run --config c5; run --config c5 --no-obs; done 2>&1 | tee gpurun_out/r03_c5_noobs.txt
run --config c3 | tee -a gpurun_out/r03_c5_noobs.txt
