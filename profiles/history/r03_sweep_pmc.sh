#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export ANTSRL_LIB=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for flat in 1 0; do
  if [ $flat = 1 ]; then export ANTSRL_SWEEP_FLAT=1; else unset ANTSRL_SWEEP_FLAT; fi
  echo "== flat=$flat"
  BENCH_ARGS="--config c4 --age 50" bash profiles/pmc_pass.sh c4_fetch_flat$flat FETCH_SIZE | grep "k_sweep\|k_perceive"
  BENCH_ARGS="--config c4 --age 50" bash profiles/pmc_pass.sh c4_tcc_flat$flat TCC_HIT_sum TCC_MISS_sum | grep "k_sweep"
done
