#!/bin/bash
# odd observations walk the environments backwards (product) against always forwards (variant -DANTSRL_ENV_ORDER_FORWARD)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/antsrl_amd/lib/variants/fwdorder.so
for args in "--config c3" "--config c3 --envs 768" "--config c5" "--config c2" "--config c4"; do for rep in 1 2; do for lib in $R/antsrl_amd/lib/libantsrl_hip.so $V; do
env ANTSRL_LIB=$lib python bench.py $args --warmup 400 --no-cpu-baseline --no-explicit-sweep --repeats 2 --steps 200 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-24s %-18s ms/step %.4f  %s' % ('$args', '$(basename $lib)', d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done; done
