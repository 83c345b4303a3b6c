#!/bin/bash
# c5 with plain (cached) observation stores (variant -DANTSRL_STORE_PLAIN): do the rows stay in the Infinity Cache for the policy?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
P=$R/antsrl_amd/lib/variants/plain_prof.so
for rep in 1 2; do for v in A=1 ANTSRL_POLICY_NEWEST_FIRST=1; do
env ANTSRL_LIB=$P $v python bench.py --config c5 --no-cpu-baseline --no-explicit-sweep --repeats 3 --steps 200 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('c5 plain stores %-30s ms/step %.4f  %.3e ant-steps/s  %s' % ('$v', d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))"
done; done
