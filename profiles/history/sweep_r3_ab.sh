#!/bin/bash
# c4 (512x512 grid, radius-3 Gaussian, rank 1): the two-columns-per-lane separable stencil against the one-column march
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for i in 1 2; do for v in "A=1" "ANTSRL_SWEEP_ONE_COLUMN=1"; do
  env ANTSRL_LIB=$P $v python bench.py --config c4 --no-cpu-baseline --no-explicit-sweep --repeats 2 --steps 50 --warmup 10 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; sw=[v for n,v in k.items() if 'sweep' in n][0]; print('%-28s ms/step=%.4f %s  sweep %.2f TB/s' % ('$v', d['ms_per_step'], k, 1024*(2*2*512*512*4+512*512/8)/sw/1e9))"
done; done
