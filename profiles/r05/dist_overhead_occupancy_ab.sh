R=${GRAFT_REPO_ROOT:-/root/repo}
run() { name=$1; shift
  env "$@" python3 $R/bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-explicit-sweep 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-22s ms/step=%.4f %s gather_overhead_us=%s' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms'], d.get('gather_overhead_us')))"; }
for i in 1 2; do
  run cur_none X=1
  run cur_dist ANTSRL_BENCH_FORCE_DIST=1
  run nocap_none ANTSRL_LIB=$R/antsrl_amd/lib/variants/nocap.so
  run nocap_dist ANTSRL_LIB=$R/antsrl_amd/lib/variants/nocap.so ANTSRL_BENCH_FORCE_DIST=1
done
