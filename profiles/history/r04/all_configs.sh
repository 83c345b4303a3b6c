#!/bin/bash
# every bench configuration on one box (bench.py's defaults: episode age 400, tune_placement), one line each
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in "c3" "c2" "c4" "c5" "c5 --no-obs" "c3 --diffuse 0.02" "c1"; do
  python3 $R/bench.py --config $c --steps 100 --repeats 3 --no-cpu-baseline --no-explicit-sweep 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-18s ms/step=%.4f value=%.3e' % ('$c', d['ms_per_step'], d['value']), r.get('kernel_ms'), 'frac=%s step_frac=%s' % (r.get('frac'), r.get('step_frac')), 'placement', d['config'].get('placement_trials_ms_per_step'))"
done
