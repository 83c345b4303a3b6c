#!/bin/bash
# in-loop policy: W1's first fragments fetched in front of the workgroup's last barrier, the fp32 constants through LDS (polpre) against before
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -x -q -m gpu -k "policy or rlapi or contract" > gpurun_out/r03_polpre_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r03_polpre_tests.log
[ $rc -eq 0 ] || exit 1
{
for cfg in "--config c5" "--config c5 --no-obs"; do
  echo "# $cfg"; VARIANTS="before polpre" ROUNDS=3 bash profiles/abn.sh $cfg
done
} | tee gpurun_out/r03_polpre_ab.txt
