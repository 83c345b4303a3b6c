#!/bin/bash
# k_act plan for batches that do not fill the chip: default choice (1024 threads when E <= CUs/2 and N >= 512)
# against the pinned 512-thread plan (ANTSRL_ACT_PLAN=1) and the pinned 1024-thread plan (=4)
R=${GRAFT_REPO_ROOT:-/root/repo}
for e in 1 16 64 128 192 256; do for plan in -1 1 4; do
ANTSRL_ACT_PLAN=$plan python3 $R/bench.py --config c3 --envs $e --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('c3 E=$e plan=$plan ms/step=%.4f %s' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
