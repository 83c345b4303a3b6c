#!/bin/bash
# On a slow box (k_perceive ~0.197 at c3): does another shape of the write comb recover the fast state?
# ants per wave (ANTSRL_PRC_RUN, profiling library) from 4 to 16; stops early on a fast box.
R=${GRAFT_REPO_ROOT:-/root/repo}
b() { python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline ${@} 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('ms/step=%.4f' % d['ms_per_step'], d['roofline']['kernel_ms'])"; }
echo "== $(date +%H:%M:%S)"
first=$(b)
echo "product            $first"
kp=$(echo "$first" | sed -n "s/.*'k_perceive': \([0-9.]*\).*/\1/p")
export ANTSRL_LIB=$R/antsrl_amd/lib/libantsrl_hip_prof.so
if python3 -c "import sys; sys.exit(0 if float('${kp:-0}') > 0.185 else 1)"; then echo "   slow box"; else echo "   fast box"; fi
for run in 8 4 5 6 12 16; do echo -n "prof PRC_RUN=$run     "; ANTSRL_PRC_RUN=$run b; done
echo -n "prof NO_TILED       "; ANTSRL_NO_TILED=1 b
echo -n "prof 1000 envs      "; b --envs 1000
echo -n "prof 1016 envs      "; b --envs 1016
