#!/bin/bash
# every bench configuration, product path (k_move + k_perceive) against round 1's k_act, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for c in ${CONFIGS:-c2 c3 c4 c5 c1}; do
  for mode in new legacy; do
    if [ $mode = legacy ]; then ACT="--act-path kact"; else ACT="--act-path meta"; fi
    timeout -k 10 280 python bench.py --config $c $ACT --no-cpu-baseline --no-explicit-sweep --repeats 3 --steps 200 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-3s %-7s ms/step=%.4f  %.3e ant-steps/s  %s' % ('$c', '$mode', d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))" || echo "$c $mode FAILED"
  done
done
