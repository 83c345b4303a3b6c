#!/bin/bash
# c5: the DQN net inside k_perceive (antsrl_set_inloop_policy) against its own kernel over the observation tensor
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do for v in inloop separate; do
python bench.py --config c5 --policy-kernel $v --no-cpu-baseline --no-explicit-sweep --repeats 3 --steps 200 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('c5 %-9s ms/step %.4f  %.3e ant-steps/s  %s' % ('$v', d['ms_per_step'], d['value'], d['roofline']['kernel_ms']))"
done; done
