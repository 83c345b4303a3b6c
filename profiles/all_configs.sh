#!/bin/bash
# every bench configuration on one box (results: one JSON line each)
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in c1 c2 c3 c4 c5; do
  python3 $R/bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('$c', 'ms/step=%.4f' % d['ms_per_step'], 'value=%.3e' % d['value'], d['roofline'].get('kernel_ms'), 'frac=%.3f' % d['roofline']['frac'])"
done
