#!/bin/bash
# Where the kernel arguments live: HIP_FORCE_DEV_KERNARG=0 (host memory, fetched over PCIe by the first waves) against =1 (device
# memory) against the runtime's default.  Same device, alternating; c3 and c2 (the latency-bound one).
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { name=$1; shift
  env "$@" python3 $R/bench.py --steps 100 --warmup 5 --repeats 3 --no-cpu-baseline --no-explicit-sweep $CFG 2>/dev/null | python3 -c "
import sys, json; d=json.loads(sys.stdin.read()); print('%-10s ms/step=%.4f %s' % ('$name', d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for CFG in "" "--config c2"; do
  echo "== ${CFG:-c3}"
  for i in 1 2; do
    run default X=1
    run dev HIP_FORCE_DEV_KERNARG=1
    run host HIP_FORCE_DEV_KERNARG=0
  done
done
