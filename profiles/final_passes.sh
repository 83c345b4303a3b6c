#!/bin/bash
# The round's profile set, one gpurun call:  bash profiles/final_passes.sh
# kernel trace + stats of the default bench, then PMC passes (each in its own run, --kernel-trace only).
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- \
    python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/prof_final.err
bash $R/profiles/pmc_pass.sh f_fetch FETCH_SIZE
bash $R/profiles/pmc_pass.sh f_write WRITE_SIZE
bash $R/profiles/pmc_pass.sh f_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY
bash $R/profiles/pmc_pass.sh f_tcc TCC_HIT_sum TCC_MISS_sum   # (more TCC counters in one pass exceed the hardware's counter slots: rocprofv3 aborts)
BENCH_ARGS=--explicit-sweep bash $R/profiles/pmc_pass.sh x_fetch FETCH_SIZE
BENCH_ARGS=--explicit-sweep bash $R/profiles/pmc_pass.sh x_write WRITE_SIZE
cd $R && python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 1500 gpurun_out/bench_final.json
