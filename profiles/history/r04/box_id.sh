#!/bin/bash
# Which physical GPU is this box, and how fast is c3 on it?  (One line per gpurun call: are the slow "boxes" the same devices?)
R=${GRAFT_REPO_ROOT:-/root/repo}
id=$(rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -n 1 | sed 's/.*: *//')
bus=$(rocm-smi --showbus 2>/dev/null | grep "GPU\[" | head -n 1 | sed 's/.*: *//')
host=$(cat /proc/sys/kernel/random/boot_id 2>/dev/null | cut -c1-8)
vb=$(rocm-smi --showvbios 2>/dev/null | grep "GPU\[" | head -n 1 | sed 's/.*: *//')
res=$(python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('ms/step=%.4f' % d['ms_per_step'], d['roofline']['kernel_ms'], 'tuner', (d['config'].get('placement_trials_ms_per_step') or {}).get('ms_per_step'), (d['config'].get('placement_trials_ms_per_step') or {}).get('chosen'))")
echo "$(date +%H:%M:%S) uid=$id bus=$bus boot=$host vbios=$vb  $res"
