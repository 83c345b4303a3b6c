#!/bin/bash
# the update deferred under an explicit sweep (k_update_move, then k_perceive, then the step's sweep) against the immediate
# form (sweep, k_move, k_perceive, k_update_one): profiling library, ANTSRL_NO_DEFER_UPDATE, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
export ANTSRL_LIB=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for c in "c4" "c3 --diffuse 0.02"; do for i in 1 2 3; do for v in "X=1" "ANTSRL_NO_DEFER_UPDATE=1"; do
  env $v python3 $R/bench.py --config $c --steps 100 --repeats 3 --no-cpu-baseline --no-explicit-sweep 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-18s %-26s %.4f ms/step' % ('$c', '$v', d['ms_per_step']), d['roofline']['kernel_ms'])"
done; done; done
