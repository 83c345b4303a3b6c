#!/bin/bash
# is k_perceive latency-bound now?  workgroups per CU through an LDS pad (profiling library): c3 7 -> 6 / 5 / 4 / 3, c5 act-only 6 -> 5 / 4 / 3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { tag=$1; shift; "$@" 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-10s ms/step=%.4f %s' % ('$tag', d['ms_per_step'], d['roofline']['kernel_ms']))"; }
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-explicit-sweep"
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
{
echo "# c3 (13 KB of LDS per workgroup; 7 workgroups per CU by VGPRs)"
for pad in 0 9 13 19 27 40; do run pad$pad env ANTSRL_LIB=$P ANTSRL_PRC_LDS_PAD=$pad $B --config c3; done
echo "# c5 act-only (24 KB; 6 per CU)"
for pad in 0 3 8 16 29; do run pad$pad env ANTSRL_LIB=$P ANTSRL_PRC_LDS_PAD=$pad $B --config c5 --no-obs; done
echo "# c2"
for pad in 0 9 19 40; do run pad$pad env ANTSRL_LIB=$P ANTSRL_PRC_LDS_PAD=$pad $B --config c2; done
} | tee gpurun_out/r03_occ_now.txt
