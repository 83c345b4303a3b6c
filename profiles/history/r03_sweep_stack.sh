#!/bin/bash
# k_sweep_sep2: four stacked segments per workgroup marching away from / towards their shared boundaries (shipped) against
# the flat (strip, segment) packing (ANTSRL_SWEEP_FLAT=1, profiling library): parity, then c4 A/B on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stencil or radius3 or golden or config4" > gpurun_out/r03_sweep_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_sweep_tests.log
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for i in 1 2 3; do
  for flat in 1 0; do
    if [ $flat = 1 ]; then export ANTSRL_SWEEP_FLAT=1; else unset ANTSRL_SWEEP_FLAT; fi
    ANTSRL_LIB=$P python3 bench.py --config c4 --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('flat=$flat ms/step=%.4f %s' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
unset ANTSRL_SWEEP_FLAT
