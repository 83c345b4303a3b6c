#!/bin/bash
# three default bench runs (separate processes) on whatever box this call lands on: the spread of the headline
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do python3 $R/bench.py --no-cpu-baseline --no-explicit-sweep --steps 100 --repeats 3 ${EXTRA} 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step' % d['ms_per_step'], d['roofline']['kernel_ms'], 'frac %.3f step_frac %.3f' % (d['roofline']['frac'], d['roofline']['step_frac']))"; done
