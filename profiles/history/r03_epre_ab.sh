#!/bin/bash
# the epilogue's loads (holding, seed): cur = issued in front of the loop; epre2 = in front of the prologue's barrier, and no load at all on the
# exploration reward's path.  tests, then same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/r03_epre_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r03_epre_tests.log
[ $rc -eq 0 ] || exit 1
{
for cfg in "--config c3" "--config c2" "--config c5" "--config c5 --no-obs" "--config c4 --steps 50"; do
  echo "# $cfg"; VARIANTS="cur epre2" ROUNDS=3 bash profiles/abn.sh $cfg
done
} | tee gpurun_out/r03_epre2_ab.txt
