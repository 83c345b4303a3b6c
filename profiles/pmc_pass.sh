#!/bin/bash
# Usage (on the GPU box, via gpurun):  bash profiles/pmc_pass.sh <tag> <counter> [<counter> ...]
# One rocprofv3 --pmc pass (counters in their own run, with --kernel-trace only) over a short bench.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp
timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- \
    python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing ${BENCH_ARGS} > $R/gpurun_out/pmc_$tag.log 2>&1
