#!/bin/bash
# What is different on a box where k_perceive runs 0.197 instead of 0.169 ms at c3?  One record per box:
# the default placement's k_perceive, raw streams (torch fill / copy), idle HBM latency + stand-alone scattered gathers
# (lat_probe), k_perceive without its stores / without its gathers (ablation variants), temperatures, clocks.
#   bash profiles/r04/box_diag.sh > gpurun_out/r04_box_diag_$(date +%H%M%S).txt
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$R/antsrl_amd/lib/variants
b() { ANTSRL_LIB=$1 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline ${@:2} 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('ms/step=%.4f' % d['ms_per_step'], d['roofline']['kernel_ms'], 'copy %.0f GB/s' % d['roofline'].get('measured_copy_gbs', 0), 'placement', (d['config'].get('placement_trials_ms_per_step') or {}).get('ms_per_step'))"; }
echo "== $(date +%H:%M:%S) $(hostname)"
echo -n "product           "; b $R/antsrl_amd/lib/libantsrl_hip.so
echo -n "product 512 envs  "; b $R/antsrl_amd/lib/libantsrl_hip.so --envs 512
echo -n "product 256 envs  "; b $R/antsrl_amd/lib/libantsrl_hip.so --envs 256
echo -n "product 2048 envs "; b $R/antsrl_amd/lib/libantsrl_hip.so --envs 2048
echo -n "no stores (ABL 2) "; b $V/abl_nostore.so
echo -n "no gathers (ABL 1)"; b $V/abl_nogather.so
python3 $R/profiles/hbm_bw_probe.py 2>/dev/null | tr '\n' ';'; echo
$R/profiles/r04/bin/lat_probe 2>&1 | grep -v amdgpu.ids
rocm-smi --showtemp --showpower --showclocks 2>&1 | grep -i "temp\|power\|mclk\|fclk\|sclk" | head -12
echo -n "product again     "; b $R/antsrl_amd/lib/libantsrl_hip.so
