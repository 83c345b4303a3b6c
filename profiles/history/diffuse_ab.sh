#!/bin/bash
# the reference's 3x3 diffusion (DIFFUSE_FACTOR = 0.02) on c3: two columns per lane against one column per lane
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for i in 1 2; do for v in "A=1" "ANTSRL_SWEEP_ONE_COLUMN=1"; do
  env ANTSRL_LIB=$P $v python bench.py --diffuse 0.02 --no-cpu-baseline --no-explicit-sweep --repeats 2 --steps 100 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('%-28s ms/step=%.4f %s  sweep %.2f TB/s' % ('$v', d['ms_per_step'], k, 1024*(2*2*65536*4+65536)/(k.get('k_sweep_r1x2') or k.get('k_sweep_march'))/1e9))"
done; done
