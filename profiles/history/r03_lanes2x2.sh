#!/bin/bash
# lane <-> perceived-cell mapping: four consecutive lanes = a 2 x 2 sub-block of the patch (lanes2x2) against four cells of a patch row (rowlanes)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ANTSRL_LIB=$R/antsrl_amd/lib/variants/lanes2x2.so python -m pytest tests -x -q -m gpu > gpurun_out/r03_l22_tests.log 2>&1; rc=$?; echo "tests(lanes2x2) rc=$rc"; tail -3 gpurun_out/r03_l22_tests.log
{
for cfg in "--config c3" "--config c2" "--config c5" "--config c4 --steps 50"; do
  echo "# $cfg"; VARIANTS="rowlanes lanes2x2" ROUNDS=3 bash profiles/abn.sh $cfg
done
} | tee gpurun_out/r03_lanes2x2.txt
