#!/bin/bash
# k_perceive: ants per wave on the smaller batches (c2: 65 536 ants, c5: 262 144), profiling-library knob
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for c in c2 c5; do for run in 8 4 2 8 4 2; do
env ANTSRL_LIB=$P ANTSRL_PRC_RUN=$run python bench.py --config $c --no-cpu-baseline --no-explicit-sweep --repeats 2 --steps 200 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%s run %d  ms/step %.4f  %s' % ('$c', $run, d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
