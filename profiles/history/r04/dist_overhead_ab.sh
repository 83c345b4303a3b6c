run() { # name, env...
  name=$1; shift
  env "$@" ANTSRL_BENCH_FORCE_DIST=1 python bench.py --steps 100 --warmup 5 --age 100 --repeats 3 --no-cpu-baseline --no-explicit-sweep --no-kernel-timing --gather ${GATHER:-staged} > gpurun_out/dab_$name.json 2> gpurun_out/dab_$name.err || echo FAIL $name
  python - <<PY
import json
d=json.loads(open("gpurun_out/dab_$name.json").read().strip().splitlines()[-1])
print("$name", d["ms_per_step"], d["ms_per_step_regions"])
PY
}
python bench.py --steps 100 --warmup 5 --age 100 --repeats 3 --no-cpu-baseline --no-explicit-sweep --no-kernel-timing > gpurun_out/dab_none.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/dab_none.json').read().strip().splitlines()[-1]); print('none', d['ms_per_step'], d['ms_per_step_regions'])"
run staged X=1
run staged_q8 GPU_MAX_HW_QUEUES=8
run staged_q2 GPU_MAX_HW_QUEUES=2
GATHER=zero_copy run zc X=1
GATHER=zero_copy run zc_q8 GPU_MAX_HW_QUEUES=8
run staged_hp TORCH_NCCL_HIGH_PRIORITY=1
