#!/bin/bash
# N-way alternating A/B on one box:  ROUNDS=3 VARIANTS="a b c" bash profiles/abn.sh [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$R/antsrl_amd/lib/variants
for i in $(seq ${ROUNDS:-3}); do for v in $VARIANTS; do
  ANTSRL_LIB=$V/$v.so python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-explicit-sweep "$@" 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-10s ms/step=%.4f %s' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms']))" || exit 1
done; done
