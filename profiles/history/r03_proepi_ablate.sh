#!/bin/bash
# upper bound of hiding the workgroup's prologue / epilogue chains (ablations: wrong results): what would a persistent kernel with
# cross-tile prefetch gain?  nosincos: frames without the sincos; hashfr: no prologue load (ants at hashed places); noepi: no epilogue
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
for cfg in "--config c3" "--config c5" "--config c5 --no-obs" "--config c2"; do
  echo "# $cfg"; VARIANTS="cur nosincos hashfr hashfr_nosincos noepi" ROUNDS=2 bash profiles/abn.sh $cfg
done
} | tee gpurun_out/r03_proepi_ablate.txt
