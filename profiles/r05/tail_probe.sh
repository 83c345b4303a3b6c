#!/bin/bash
# VERDICT r4 item 1, "step 0" (results WRONG by design, profiling library): how much of k_update_move's latency chain does
# k_perceive's observation stream hide when the chain runs inside the k_perceive launch?
#   ANTSRL_TAIL_PROBE unset: the product's two launches (k_update_move + k_perceive)
#   1: E extra 256-thread workgroups IN FRONT of k_perceive's grid run update + move (loop forms) of env blockIdx.x; no k_update_move launch
#   2: the workgroup of every environment's LAST segment goes on to run update + move of an unrelated environment
# Alternating runs in one call (one device).  Usage on the GPU box: bash profiles/r05/tail_probe.sh [rounds] [config...]
R=${GRAFT_REPO_ROOT:-/root/repo}
export ANTSRL_LIB=$R/antsrl_amd/lib/libantsrl_hip_prof.so
N=${1:-3}; shift
for c in ${@:-c3 c2}; do
for i in $(seq $N); do for m in ${MODES:-0 1 2 3}; do
  if [ $m = 0 ]; then unset ANTSRL_TAIL_PROBE; else export ANTSRL_TAIL_PROBE=$m; fi
  python3 $R/bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --no-explicit-sweep 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('$c probe=$m ms/step=%.4f %s' % (d['ms_per_step'], d['roofline']['kernel_ms']))" || exit 1
done; done; done
