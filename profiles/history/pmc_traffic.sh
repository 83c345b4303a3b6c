#!/bin/bash
# HBM traffic of every kernel of a step (rocprofv3 --pmc, separate passes): FETCH_SIZE, WRITE_SIZE, L2 hit/miss.
# BENCH_ARGS selects the configuration (default c3).  Prints per-kernel means; profiles/summarize_traffic.py turns
# the csv files into profiles/traffic_<config>.json.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${TAG:-t}
bash $R/profiles/pmc_pass.sh ${tag}_fetch FETCH_SIZE
bash $R/profiles/pmc_pass.sh ${tag}_write WRITE_SIZE
bash $R/profiles/pmc_pass.sh ${tag}_tcc TCC_HIT_sum TCC_MISS_sum
