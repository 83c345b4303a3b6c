#!/bin/bash
# marching stencils after the prefetch over-read fix: parity, timing (c4 radius 3; c3 with the reference's 3x3 diffusion), FETCH_SIZE
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stencil or radius3 or golden or config4 or diffus" > gpurun_out/r03_sweep_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_sweep_tests.log
run() { python3 bench.py "$@" --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('ms/step=%.4f %s' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for i in 1 2; do run --config c4; run --config c3 --diffuse 0.02; done
BENCH_ARGS="--config c4 --age 50" bash profiles/pmc_pass.sh c4_fetch_fix FETCH_SIZE | grep "k_sweep"
BENCH_ARGS="--config c3 --diffuse 0.02 --age 50" bash profiles/pmc_pass.sh c3d_fetch_fix FETCH_SIZE | grep "k_sweep"
