#!/bin/bash
# A stand-alone model of k_perceive (profiles/history/obs_stream_probe.hip: its write comb + one 16-byte gather per lane and
# row from a 1 MiB window per environment of a 1 GiB table): does it show the fast-box / slow-box difference of the real kernel?
R=${GRAFT_REPO_ROOT:-/root/repo}; P=$R/profiles/r04/bin/obs_stream_probe
echo "== $(date +%H:%M:%S)"
python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('bench c3 ms/step=%.4f' % d['ms_per_step'], d['roofline']['kernel_ms'], (d['config'].get('placement_trials_ms_per_step') or {}).get('ms_per_step'))"
for i in 1 2 3; do $P --lds 13 --run 8 --map 1 --gather 1; done
for i in 1 2; do $P --lds 13 --run 8 --map 1 --gather 0; done
$P --lds 13 --run 8 --map 0 --gather 1
