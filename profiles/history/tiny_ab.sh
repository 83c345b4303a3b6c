#!/bin/bash
# small batches: the cell-meta path (k_update_move + k_perceive) against k_act + k_update_one — is the tiny-batch rule
# of AntsCfg.act_path = AUTO still right with the deferred update?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for args in "--config c1" "--config c1 --envs 8" "--config c1 --envs 64" "--config c2 --envs 16" "--config c2 --envs 32"; do for act in meta kact; do
python bench.py $args --act-path $act --no-cpu-baseline --no-explicit-sweep --no-kernel-timing --repeats 3 --steps 1000 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-26s %-5s ms/step %.4f  %.3e ant-steps/s' % ('$args', '$act', d['ms_per_step'], d['value']))"
done; done; done
