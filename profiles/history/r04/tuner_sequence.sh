R=${GRAFT_REPO_ROOT:-/root/repo}
one() { python3 $R/bench.py "$@" --steps 100 --repeats 3 --no-cpu-baseline --no-explicit-sweep 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('placement_trials_ms_per_step'))"; }
for i in 1 2 3; do one --config c2; done
for i in 1 2 3 4 5 6; do one --config c3; done
