#!/bin/bash
# Profiling ablations of k_act (results invalid while ANTSRL_ABLATE is set): per-kernel ms from bench.py.
# NOTE: 256 (no perception), 512 (no gathers) and 1024 (no stores) select the GENERIC perception loop, not the
# pipelined one the bench normally runs — compare them with each other, not with ablate=0; 2048 (no
# explored-map marks) and 32768 (phase timeline) keep the pipelined loop.
R=${GRAFT_REPO_ROOT:-/root/repo}
for a in 0 256 512 1024 2048 1536 3584; do
  ANTSRL_ABLATE=$a python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('ablate=%-5s ms/step=%.4f kernel_ms=%s' % ('$a', d['ms_per_step'], d['roofline']['kernel_ms']))"
done
