#!/bin/bash
# cell records in blocks of 2 x 4 cells per line (KP::tiled, shipped) against row-major records (ANTSRL_NO_TILED=1, profiling
# library): the GPU suite on the product library, a parity subset on the row-major path, then alternating bench runs on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/r03_tiled_tests.log 2>&1; echo "tiled tests rc=$?"; tail -3 gpurun_out/r03_tiled_tests.log
export ANTSRL_LIB=$R/antsrl_amd/lib/libantsrl_hip_prof.so
ANTSRL_NO_TILED=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_generate.py -x -q -m gpu > gpurun_out/r03_rowmajor_tests.log 2>&1; echo "row-major tests rc=$?"; tail -2 gpurun_out/r03_rowmajor_tests.log
run() { python3 bench.py "$@" --steps 200 --warmup 20 --no-cpu-baseline --no-explicit-sweep 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('$TAG $* ms/step=%.4f %s' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for i in 1 2 3; do
  export ANTSRL_NO_TILED=1; TAG="row-major"; run --config c3
  unset ANTSRL_NO_TILED; TAG="tiled    "; run --config c3
done
for c in c2 c5; do for i in 1 2; do
  export ANTSRL_NO_TILED=1; TAG="row-major"; run --config $c
  unset ANTSRL_NO_TILED; TAG="tiled    "; run --config $c
done; done
