#!/bin/bash
# Ants per wave of k_perceive (pick_run: 8 since round 2, tuned at six workgroups per CU) re-checked at seven: profiling
# library, ANTSRL_PRC_RUN = 4 / 8 / 16, alternating on one device.
R=${GRAFT_REPO_ROOT:-/root/repo}
export ANTSRL_LIB=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for c in "--config c3" "--config c2" "--config c4"; do
  echo "== $c"
  for i in 1 2; do for r in 4 8 16; do
    ANTSRL_PRC_RUN=$r python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-explicit-sweep $c 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('run %-3s ms/step=%.4f %s' % ('$r', d['ms_per_step'], d['roofline']['kernel_ms']))" || exit 1
  done; done
done
