#!/bin/bash
# What the per-step reward / done all-gather costs ONE rank (RCCL, world size 1: ANTSRL_BENCH_FORCE_DIST): same device,
# alternating.  Round 4 measured +4 us per step with the collective on a high-priority stream (bench.py's default).
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { name=$1; shift
  env "$@" python3 $R/bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-explicit-sweep --gather ${GATHER:-staged} 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-14s ms/step=%.4f %s gather_overhead_us=%s' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d.get('gather_overhead_us')))"; }
for i in 1 2; do
  run none X=1
  run staged_hp ANTSRL_BENCH_FORCE_DIST=1
  run staged_lowprio ANTSRL_BENCH_FORCE_DIST=1 TORCH_NCCL_HIGH_PRIORITY=0
  GATHER=zero_copy run zc_hp ANTSRL_BENCH_FORCE_DIST=1
  GATHER=inline run inline ANTSRL_BENCH_FORCE_DIST=1
done
