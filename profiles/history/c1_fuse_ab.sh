#!/bin/bash
# the reference's own case (c1: 1 env x 32 ants, 64x64): update fused at the tail of k_act against two launches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=$R/antsrl_amd/lib/libantsrl_hip_prof.so
for rep in 1 2 3; do for v in A=1 ANTSRL_FUSE_UPDATE=1; do
env ANTSRL_LIB=$P $v python bench.py --config c1 --no-cpu-baseline --no-explicit-sweep --repeats 3 --steps 500 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('c1 %-22s ms/step %.4f  %s' % ('$v', d['ms_per_step'], d['roofline']['kernel_ms']))"
env ANTSRL_LIB=$P $v python bench.py --config c1 --no-kernel-timing --no-cpu-baseline --no-explicit-sweep --repeats 3 --steps 500 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('c1 %-22s ms/step %.4f  (no kernel events)' % ('$v', d['ms_per_step']))"
done; done
