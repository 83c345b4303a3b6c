#!/bin/bash
# interleaved waves inside a k_perceive workgroup (variant ilv) against the shipped mapping: parity subset, then A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/antsrl_amd/lib/variants
ANTSRL_LIB=$V/ilv.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py tests/test_gpu_policy.py -x -q -m gpu > gpurun_out/r03_ilv_tests.log 2>&1; echo "ilv tests rc=$?"; tail -3 gpurun_out/r03_ilv_tests.log
bash profiles/ab.sh run base_r3 ilv 3 --no-explicit-sweep 2>&1 | tee gpurun_out/r03_ilv_ab.txt
for run in 4 16; do for v in base_r3 ilv; do
  ANTSRL_PRC_RUN=$run ANTSRL_LIB=$V/$v.so python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-explicit-sweep 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('run=$run %-8s ms/step=%.4f %s' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done 2>&1 | tee -a gpurun_out/r03_ilv_ab.txt
for v in base_r3 ilv; do
  ANTSRL_LIB=$V/$v.so python3 bench.py --config c5 --steps 200 --warmup 20 --no-cpu-baseline --no-explicit-sweep 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('c5 %-8s ms/step=%.4f %s' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms']))"
done 2>&1 | tee -a gpurun_out/r03_ilv_ab.txt
