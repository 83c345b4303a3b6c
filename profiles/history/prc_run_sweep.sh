#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
one() { local label=$1; shift
  env ANTSRL_LIB=$P "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-explicit-sweep --repeats 2 --steps 100 --warmup $WARM ${BENCH_ARGS} 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-28s ms/step=%.4f %s' % ('$label', d['ms_per_step'], d['roofline']['kernel_ms']))" || echo "$label FAILED"
}
for WARM in 400 10; do echo "warmup $WARM"; for r in 2 4 8 32; do one "run=$r" ANTSRL_PRC_RUN=$r; done; done
