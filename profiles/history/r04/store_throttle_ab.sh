# k_perceive with a cap on the vector-memory operations in flight when a group's observation stores are issued (s_waitcnt
# vmcnt(N) in front of them; a patch, not in the tree): does holding the store stream back let the gathers through on a slow box?
R=${GRAFT_REPO_ROOT:-/root/repo}; V=$R/antsrl_amd/lib/variants
b() { ANTSRL_LIB=$1 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('ms/step=%.4f' % d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for i in 1 2; do for v in thr0 thr8 thr6 thr4; do echo -n "$v  "; b $V/$v.so; done; done
