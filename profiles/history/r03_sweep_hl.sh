#!/bin/bash
# k_sweep_sep2 A/B on one box (profiling library): halo lanes 4 (whole-line stores, shipped) / minimal; flat / stacked
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stencil or radius3 or golden or config4" > gpurun_out/r03_sweep_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_sweep_tests.log
export ANTSRL_LIB=$R/antsrl_amd/lib/libantsrl_hip_prof.so
run() { python3 bench.py --config c4 --steps 100 --warmup 10 --repeats 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('$1 ms/step=%.4f %s' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for i in 1 2; do
  unset ANTSRL_SWEEP_MINHALO ANTSRL_SWEEP_STACK; run "hl4 flat "
  export ANTSRL_SWEEP_MINHALO=1; run "min flat "
  export ANTSRL_SWEEP_STACK=1; run "min stack"
  unset ANTSRL_SWEEP_MINHALO; run "hl4 stack"
  unset ANTSRL_SWEEP_STACK
done
for m in 0 1; do
  if [ $m = 1 ]; then export ANTSRL_SWEEP_MINHALO=1; else unset ANTSRL_SWEEP_MINHALO; fi
  echo "== minhalo=$m"
  BENCH_ARGS="--config c4 --age 50" bash profiles/pmc_pass.sh c4_fetch_mh$m FETCH_SIZE | grep "k_sweep"
  BENCH_ARGS="--config c4 --age 50" bash profiles/pmc_pass.sh c4_write_mh$m WRITE_SIZE | grep "k_sweep"
done
