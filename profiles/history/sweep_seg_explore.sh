#!/bin/bash
# rows per march segment, per sweep kernel (profiling library knobs): c4 (radius-3 Gaussian) two-column and one-column,
# c3 + the reference's 3x3 diffusion two-column and one-column
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
run() { cfg=$1; shift; env ANTSRL_LIB=$P "$@" python bench.py $cfg --no-cpu-baseline --no-explicit-sweep --repeats 1 --steps 30 --warmup 5 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; sw=[(n,v) for n,v in k.items() if 'sweep' in n][0]; print('%-22s %-52s %s %.4f ms' % ('$cfg', '$*', sw[0], sw[1]))"; }
for seg in 16 32 64 128; do run "--config c4" ANTSRL_SWEEP_SEG=$seg; done
for seg in 16 32 64 128; do run "--config c4" ANTSRL_SWEEP_ONE_COLUMN=1 ANTSRL_SWEEP_SEG=$seg; done
for seg in 16 32 64 128; do run "--diffuse 0.02" ANTSRL_SWEEP_SEG=$seg; done
for seg in 16 32 64 128; do run "--diffuse 0.02" ANTSRL_SWEEP_ONE_COLUMN=1 ANTSRL_SWEEP_SEG=$seg; done
