#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
one() { local label=$1; shift
  env ANTSRL_LIB=$P "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-explicit-sweep --repeats 2 --steps 100 --warmup 400 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-28s ms/step=%.4f %s' % ('$label', d['ms_per_step'], d['roofline']['kernel_ms']))" || echo "$label FAILED"
}
one "pad=0" A=1
one "pad=20 (5 WG/CU)" ANTSRL_PRC_LDS_PAD=20
one "pad=40 (2 WG/CU)" ANTSRL_PRC_LDS_PAD=40
one "run=8" ANTSRL_PRC_RUN=8
one "run=16" ANTSRL_PRC_RUN=16
one "run=64" ANTSRL_PRC_RUN=64
one "pad=0 again" A=1
