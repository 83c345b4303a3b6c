#!/bin/bash
# One gpurun call: the GPU parity suite, then the default bench with the product library and, for an A/B on
# the same box, the round-1 k_act path (bench.py --act-path kact = AntsCfg.act_path ANTSRL_ACT_SINGLE_KERNEL).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -15 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-explicit-sweep > gpurun_out/bench_new_$i.json 2> gpurun_out/bench_new_$i.err || { tail -5 gpurun_out/bench_new_$i.err; exit 1; }
  timeout -k 10 300 python bench.py --act-path kact --no-cpu-baseline --no-explicit-sweep > gpurun_out/bench_legacy_$i.json 2> gpurun_out/bench_legacy_$i.err || { tail -5 gpurun_out/bench_legacy_$i.err; exit 1; }
done
python - <<'PY'
import json
for n in ("new_1", "legacy_1", "new_2", "legacy_2"):
    d = json.load(open("gpurun_out/bench_%s.json" % n))
    print(n, "ms/step %.4f" % d["ms_per_step"], d["ms_per_step_regions"], d["roofline"]["kernel_ms"], d["config"]["kernels"])
PY
