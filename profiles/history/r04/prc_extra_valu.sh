# Is k_perceive's VALU work on the critical path?  Variants with PRC_EXTRA_VALU = 64 / 128 / 256 more v_fma_f32 per 2-ant group
# (a patch of the process lambda, not in the tree: four independent chains in front of the group's work).
R=${GRAFT_REPO_ROOT:-/root/repo}; V=$R/antsrl_amd/lib/variants
b() { ANTSRL_LIB=$1 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline ${@:2} 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('ms/step=%.4f' % d['ms_per_step'], d['roofline']['kernel_ms'])"; }
for i in 1 2; do
echo -n "product      "; b $R/antsrl_amd/lib/libantsrl_hip_prof.so
echo -n "+64 VALU     "; b $V/xv64.so
echo -n "+128 VALU    "; b $V/xv128.so
echo -n "+256 VALU    "; b $V/xv256.so
done
