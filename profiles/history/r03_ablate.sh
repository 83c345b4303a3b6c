#!/bin/bash
# k_perceive ablation matrix at the shipped settings (run 8), steady state (age 400), one box, two rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/antsrl_amd/lib/variants
for i in 1 2; do for v in ${VARIANTS:-base_r3 ab_A ab_B ab_C ab_D}; do
  ANTSRL_LIB=$V/$v.so python3 bench.py --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline --no-explicit-sweep ${BENCH_ARGS} 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-10s ms/step=%.4f %s' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
