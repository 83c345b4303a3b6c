#!/bin/bash
# HBM traffic in the late regime of an episode (400 warm-up steps), product path and round-1 k_act
R=${GRAFT_REPO_ROOT:-/root/repo}
export BENCH_ARGS="--warmup 400"
bash $R/profiles/pmc_pass.sh l_fetch FETCH_SIZE | grep "k_perceive\|k_move\|k_update\|k_act"
bash $R/profiles/pmc_pass.sh l_write WRITE_SIZE | grep "k_perceive\|k_move\|k_update\|k_act"
export BENCH_ARGS="--warmup 400 --act-path kact"
bash $R/profiles/pmc_pass.sh ll_fetch FETCH_SIZE | grep "k_perceive\|k_move\|k_update\|k_act"
bash $R/profiles/pmc_pass.sh ll_write WRITE_SIZE | grep "k_perceive\|k_move\|k_update\|k_act"
