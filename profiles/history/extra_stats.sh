#!/bin/bash
# rocprofv3 kernel stats of the non-default bench configurations (c2, c4, c5), one run each
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for c in c2 c4 c5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$c -- \
      python3 $R/bench.py --config $c --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/bench_${c}_rocprof.json 2> $R/gpurun_out/prof_$c.err
done
