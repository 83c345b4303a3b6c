#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/antsrl_amd/lib/variants
for v in depth2 depth3; do ANTSRL_LIB=$V/$v.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_guard.py tests/test_gpu_policy.py -x -q -m gpu > gpurun_out/r03_${v}_tests.log 2>&1; echo "$v tests rc=$?"; tail -2 gpurun_out/r03_${v}_tests.log; done
VARIANTS="base_r3 depth2 depth3" ROUNDS=3 bash profiles/abn.sh 2>&1 | tee gpurun_out/r03_depth_ab.txt
for c in c2 c4 c5; do VARIANTS="base_r3 depth2 depth3" ROUNDS=2 bash profiles/abn.sh --config $c 2>&1 | sed "s/^/$c /" | tee -a gpurun_out/r03_depth_ab.txt; done
