#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export ANTSRL_LIB=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for i in 1 2; do for seg in 32 48 64 128; do
  ANTSRL_SWEEP_SEG=$seg python3 bench.py --config c4 --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('seg=$seg ms/step=%.4f %s' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
for i in 1 2; do for seg in 16 32 64; do
  ANTSRL_SWEEP_SEG=$seg python3 bench.py --config c3 --diffuse 0.02 --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('r1 seg=$seg ms/step=%.4f %s' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
