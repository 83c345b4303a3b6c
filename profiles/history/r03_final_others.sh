#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
CONFIGS="${CONFIGS:-c2 c5}" bash profiles/final_passes.sh 2>&1 | tee gpurun_out/r03_final_${TAG:-c2c5}.txt
