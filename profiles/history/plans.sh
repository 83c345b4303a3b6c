R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for pl in 0 2 4 5; do
ANTSRL_ACT_PLAN=$pl python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('plan=$pl ms/step=%.4f %s' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
