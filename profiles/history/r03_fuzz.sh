#!/bin/bash
# a fuzz campaign on the final tree of the round (the cell-meta test: tiled records on every second case; the general test)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ANTSRL_FUZZ_BASE=5000 ANTSRL_FUZZ_CASES=12000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "library_jitter" > gpurun_out/r03_fuzz_meta.log 2>&1; echo "meta rc=$?"; tail -3 gpurun_out/r03_fuzz_meta.log

