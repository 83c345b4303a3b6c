#!/bin/bash
# k_update_move: x / y / theta / prev handed from the update to the move in registers (fwd) against re-loaded (nofwd): tests, fuzz, same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/r03_umf_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r03_umf_tests.log
[ $rc -eq 0 ] || exit 1
ANTSRL_FUZZ_BASE=60000 ANTSRL_FUZZ_CASES=1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r03_umf_fuzz.log 2>&1; rc=$?; echo "fuzz rc=$rc"; tail -2 gpurun_out/r03_umf_fuzz.log
[ $rc -eq 0 ] || exit 1
{
for cfg in "--config c2" "--config c1" "--config c5" "--config c3"; do
  echo "# $cfg"; VARIANTS="nofwd fwd" ROUNDS=3 bash profiles/abn.sh $cfg
done
} | tee gpurun_out/r03_um_forward.txt
