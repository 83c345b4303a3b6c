R=${GRAFT_REPO_ROOT:-/root/repo}; V=$R/antsrl_amd/lib/variants
b() { ANTSRL_LIB=$1 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline ${@:2} 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('ms/step=%.4f' % d['ms_per_step'], d['roofline']['kernel_ms'])"; }
echo -n "product                "; b $R/antsrl_amd/lib/libantsrl_hip.so
echo -n "no stores (ABL 2)      "; b $V/abl_nostore.so
echo -n "no gathers (ABL 1)     "; b $V/abl_nogather.so
echo -n "neither (ABL 3)        "; b $V/abl_both.so
python3 $R/profiles/prc_trace.py c3 2>&1 | grep -v amdgpu.ids
