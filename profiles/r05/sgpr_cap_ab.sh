#!/bin/bash
# k_perceive's REAL residency: the kernel's 102-106 SGPRs admit 6 workgroups of 256 threads per CU, not the 7 its VGPR count
# suggests (MI355X_MICROARCH.md: blocks per CU = floor(800 / (ceil(sgpr / 16) * 16 + 16)): <= 80 SGPRs -> 8, 82-96 -> 7, 98+ -> 6).
# Variants: base (no cap), sg96w7 (amdgpu_num_sgpr(96), launch_bounds(256, 7)), sg96w4 (cap 96 only), sg80w8 (cap 80, bounds (256, 8)).
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$R/antsrl_amd/lib/variants
for c in "$@"; do
  echo "== $c"
  for i in 1 2; do for v in base sg96w7 sg96w4 sg80w8; do
    ANTSRL_LIB=$V/$v.so python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-explicit-sweep $c 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-8s ms/step=%.4f %s' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms']))" || exit 1
  done; done
done
