#!/bin/bash
# k_act duration against batch size (c3 shape): exposes workgroup-slot quantisation (E vs resident WGs)
R=${GRAFT_REPO_ROOT:-/root/repo}
for e in 256 512 768 1024 1536 2048 2304 3072; do
  for a in 0 4096; do
  ANTSRL_ABLATE=$a python3 $R/bench.py --envs $e --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']['k_act']; print('E=%-5s ablate=%-5s ms/step=%.4f k_act=%.4f  ns/env=%.1f' % ('$e', '$a', d['ms_per_step'], k, k*1e6/$e))"
  done
done
