R=${GRAFT_REPO_ROOT:-/root/repo}
for c in c1 c2; do for f in 0 1 0 1; do
ANTSRL_FUSE_UPDATE=$f python3 $R/bench.py --config $c --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('$c fuse=$f ms/step=%.4f %s' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
