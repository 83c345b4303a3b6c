#!/bin/bash
# c4 radius-3 sweep: rows per segment and workgroup order (profiling library knobs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
run() { env ANTSRL_LIB=$P "$@" python bench.py --config c4 --no-cpu-baseline --no-explicit-sweep --repeats 1 --steps 30 --warmup 5 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; sw=[v for n,v in k.items() if 'sweep' in n][0]; print('%-50s sweep %.4f ms %.2f TB/s' % ('$*', sw, 1024*(2*2*512*512*4+512*512/8)/sw/1e9))"; }
run ANTSRL_SWEEP_ONE_COLUMN=1
for seg in 64 48 32 24 16 32 64; do run ANTSRL_SWEEP_SEG=$seg; done
