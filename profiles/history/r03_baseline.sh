#!/bin/bash
# round-3 baseline on one box: GPU suite, then the default bench with driver flags, then the long form
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/r03_tests0.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_tests0.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_driver0.json 2> gpurun_out/r03_bench_driver0.err && tail -c 1500 gpurun_out/r03_bench_driver0.json
python bench.py --no-cpu-baseline --no-explicit-sweep > gpurun_out/r03_bench_default0.json 2>&1 && tail -c 1200 gpurun_out/r03_bench_default0.json
python bench.py --no-cpu-baseline --no-explicit-sweep --age 0 --steps 20 --warmup 5 > gpurun_out/r03_bench_age0.json 2>&1 && tail -c 1200 gpurun_out/r03_bench_age0.json
