#!/bin/bash
# k_perceive: waves per workgroup (compile-time PRC_TPB), early and late in the episode, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/antsrl_amd/lib/variants
for rep in 1 2; do
for lib in $R/antsrl_amd/lib/libantsrl_hip.so $V/tpb128.so $V/tpb64.so; do
  for wu in 20 400; do
  env ANTSRL_LIB=$lib python bench.py --no-cpu-baseline --no-explicit-sweep --repeats 1 --steps 200 --warmup $wu 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print('%-28s warmup %3d  ms/step=%.4f  k_perceive %.4f' % ('$(basename $lib)', $wu, d['ms_per_step'], k['k_perceive']))"
  done
done; done
