#!/bin/bash
# k_update_move: latency- or bandwidth-bound?  c3 with fewer envs, and without rocks
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for args in "--envs 1024" "--envs 512" "--envs 256" "--envs 128" "--envs 1024 --rocks 0" "--envs 256 --rocks 0"; do
python bench.py $args --no-cpu-baseline --no-explicit-sweep --repeats 2 --steps 200 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-26s ms/step %.4f  %s' % ('$args', d['ms_per_step'], d['roofline']['kernel_ms']))"
done
