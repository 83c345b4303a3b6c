#!/bin/bash
# deferred update (k_update_move) against update and move as two launches, same box (profiling-library switch)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for rep in 1 2; do for c in ${CONFIGS:-c3 c2 c5 c1}; do for v in A=1 ANTSRL_NO_DEFER_UPDATE=1; do
env ANTSRL_LIB=$P $v python bench.py --config $c --no-cpu-baseline --no-explicit-sweep --repeats 3 --steps 200 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-3s %-26s ms/step %.4f  %.3e ant-steps/s  %s' % ('$c', '$v', d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))"
done; done; done
