#!/bin/bash
# k_update_move: does it matter that all 1024 workgroups are resident at once and run their phases in lockstep?
# dummy dynamic LDS limits the workgroups per CU (17 KB own: 4 per CU by waves; +24 KB: 3; +40 KB: 2; +100 KB: 1)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export ANTSRL_LIB=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for i in 1 2; do for pad in 0 24 40 100; do
  ANTSRL_UM_LDS_PAD=$pad python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-explicit-sweep 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('pad=$pad ms/step=%.4f %s' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
