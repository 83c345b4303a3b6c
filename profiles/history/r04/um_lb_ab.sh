R=${GRAFT_REPO_ROOT:-/root/repo}; V=$R/antsrl_amd/lib/variants
run() { ANTSRL_LIB=$V/$1.so python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline "${@:2}" 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-8s ms/step=%.4f %s' % ('$1', d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for cfg in c3 c2 c5; do echo "# --config $cfg"; for i in 1 2; do for v in um_base um_lb8 um_lb7 um_lb6; do run $v --config $cfg || exit 1; done; done; done
