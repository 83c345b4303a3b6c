#!/bin/bash
# end-of-round validation on the final tree: every alternate code path, a fuzz campaign, the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tests/alt_paths.sh > gpurun_out/r03_alt_paths_final.log 2>&1; echo "alt_paths rc=$?"; grep -c passed gpurun_out/r03_alt_paths_final.log; grep -i "failed\|error" gpurun_out/r03_alt_paths_final.log | head
ANTSRL_FUZZ_BASE=40000 ANTSRL_FUZZ_CASES=4000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r03_fuzz_final.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r03_fuzz_final.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_flags.json 2> gpurun_out/r03_bench_driver_flags.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r03_bench_driver_flags.json
