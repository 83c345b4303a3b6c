#!/bin/bash
# k_move / k_perceive decomposition on one box: run-length sweep (profiling library, ANTSRL_PRC_RUN) and
# compile-time ablation variants (profiles/build_variants.sh).  Ablated variants compute wrong results by design.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/antsrl_amd/lib/variants
one() { # label, lib, env assignments...
  local label=$1 lib=$2; shift 2
  env ANTSRL_LIB=$lib "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-explicit-sweep --repeats 2 --steps 100 ${BENCH_ARGS} 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-28s ms/step=%.4f %s' % ('$label', d['ms_per_step'], d['roofline']['kernel_ms']))" || echo "$label FAILED"
}
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for run in ${RUNS:-8 16 32 64}; do one "run=$run" $P ANTSRL_PRC_RUN=$run; done
one "product" $R/antsrl_amd/lib/libantsrl_hip.so A=1
for v in ${VARIANTS:-nostore nomark nostore_nomark}; do one "$v" $V/$v.so A=1; done
BENCH_ARGS="$BENCH_ARGS --act-path kact" one "legacy k_act" $P A=1
