R=${GRAFT_REPO_ROOT:-/root/repo}
V=$R/antsrl_amd/lib/variants
for c in "--config c1" "--config c2 --envs 16" "--config c3 --envs 64"; do
  echo "== $c"
  for i in 1 2; do for v in cur nocap cap96w4 cap104w7; do
    ANTSRL_LIB=$V/$v.so python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-explicit-sweep $c 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-9s ms/step=%.4f %s' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms']))" || exit 1
  done; done
done
