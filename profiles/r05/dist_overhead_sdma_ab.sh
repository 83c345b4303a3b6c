R=${GRAFT_REPO_ROOT:-/root/repo}
run() { name=$1; shift
  env "$@" python3 $R/bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-explicit-sweep --gather ${GATHER:-staged} 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-22s ms/step=%.4f %s gather_overhead_us=%s' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d.get('gather_overhead_us')))"; }
run none X=1
run staged ANTSRL_BENCH_FORCE_DIST=1
run staged_nosdma ANTSRL_BENCH_FORCE_DIST=1 HSA_ENABLE_SDMA=0
GATHER=zero_copy run zc ANTSRL_BENCH_FORCE_DIST=1
GATHER=zero_copy run zc_nosdma ANTSRL_BENCH_FORCE_DIST=1 HSA_ENABLE_SDMA=0
run none_nosdma HSA_ENABLE_SDMA=0
