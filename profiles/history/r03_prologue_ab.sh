#!/bin/bash
# the prologue by one wave per workgroup (product build of the tree) against the tree before (variants/base.so): tests, fuzz, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/r03_pro_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r03_pro_tests.log
[ $rc -eq 0 ] || exit 1
run() { tag=$1; shift; "$@" 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-10s ms/step=%.4f %s' % ('$tag', d['ms_per_step'], d['roofline']['kernel_ms']))"; }
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-explicit-sweep"
V=$R/antsrl_amd/lib/variants
{
for cfg in "--config c3" "--config c2" "--config c5" "--config c5 --no-obs" "--config c4 --steps 50" "--config c3 --diffuse 0.02" "--config c1"; do
  echo "# $cfg"
  for i in 1 2 3; do
    run before env ANTSRL_LIB=$V/base.so $B $cfg
    run after $B $cfg
  done
done
} | tee gpurun_out/r03_prologue_ab.txt
