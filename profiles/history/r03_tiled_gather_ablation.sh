#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash profiles/ab.sh run base_r3 tiledg 3 --no-explicit-sweep 2>&1 | tee gpurun_out/r03_tiledg_ab.txt
for c in c2 c5; do bash profiles/ab.sh run base_r3 tiledg 2 --no-explicit-sweep --config $c 2>&1 | sed "s/^/$c /" | tee -a gpurun_out/r03_tiledg_ab.txt; done
