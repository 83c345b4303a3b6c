#!/bin/bash
# Variants in the LATE regime of an episode (ants dispersed): warm-up 400 steps, then 2 x 100 timed steps.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/antsrl_amd/lib/variants
one() { local label=$1 lib=$2; shift 2
  env ANTSRL_LIB=$lib "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-explicit-sweep --repeats 2 --steps 100 --warmup ${WARM:-400} ${BENCH_ARGS} 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-28s ms/step=%.4f %s' % ('$label', d['ms_per_step'], d['roofline']['kernel_ms']))" || echo "$label FAILED"
}
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
one "product" $R/antsrl_amd/lib/libantsrl_hip.so A=1
for v in ${VARIANTS}; do one "$v" $V/$v.so A=1; done
one "product again" $R/antsrl_amd/lib/libantsrl_hip.so A=1
BENCH_ARGS="$BENCH_ARGS --act-path kact" one "legacy k_act" $P A=1
