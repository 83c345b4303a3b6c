run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 100 --warmup 5 --age 100 --repeats 3 --no-cpu-baseline --no-explicit-sweep --no-kernel-timing --gather ${GATHER:-staged} > gpurun_out/dab_$name.json 2> gpurun_out/dab_$name.err || echo FAIL $name
  python - <<PY
import json
d=json.loads(open("gpurun_out/dab_$name.json").read().strip().splitlines()[-1])
print("$name", d["ms_per_step"], d["ms_per_step_regions"])
PY
}
run none X=1
run none_q8 GPU_MAX_HW_QUEUES=8
run staged_hp ANTSRL_BENCH_FORCE_DIST=1 TORCH_NCCL_HIGH_PRIORITY=1
GATHER=zero_copy run zc_hp ANTSRL_BENCH_FORCE_DIST=1 TORCH_NCCL_HIGH_PRIORITY=1
run staged_hp_q8 ANTSRL_BENCH_FORCE_DIST=1 TORCH_NCCL_HIGH_PRIORITY=1 GPU_MAX_HW_QUEUES=8
GATHER=zero_copy run zc_hp_q8 ANTSRL_BENCH_FORCE_DIST=1 TORCH_NCCL_HIGH_PRIORITY=1 GPU_MAX_HW_QUEUES=8
GATHER=zero_copy run zc_q8 ANTSRL_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=8
run none2 X=1
