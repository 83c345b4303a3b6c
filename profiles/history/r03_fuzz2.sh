#!/bin/bash
# after the prologue / epilogue changes of k_perceive: the k_perceive-related alternate paths, then a fuzz campaign
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ALT_PATHS="ANTSRL_PRC_RUN=32 ANTSRL_PRC_RUN=12 ANTSRL_PRC_RUN=5 ANTSRL_NO_TILED=1 ANTSRL_NO_INTERLEAVE=1 ANTSRL_NO_DEFER_UPDATE=1" bash tests/alt_paths.sh > gpurun_out/r03_alt_paths2.log 2>&1; echo "alt rc=$?"; cat gpurun_out/r03_alt_paths2.log
ANTSRL_FUZZ_BASE=20000 ANTSRL_FUZZ_CASES=5000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r03_fuzz2.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r03_fuzz2.log
