#!/bin/bash
# c5 act-only (--no-obs): where does k_perceive's time go without an output stream?  compile-time ablations + SQ counters
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
VARIANTS="base abA abM abS abP abAP abAMSP abE" ROUNDS=2 bash profiles/abn.sh --config c5 --no-obs | tee gpurun_out/r03_c5_ablate.txt
BENCH_ARGS="--config c5 --no-obs" bash profiles/pmc_pass.sh c5a_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD | grep "k_update_move\|k_perceive" | tee -a gpurun_out/r03_c5_ablate.txt
BENCH_ARGS="--config c5 --no-obs" bash profiles/pmc_pass.sh c5a_lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA | grep "k_update_move\|k_perceive" | tee -a gpurun_out/r03_c5_ablate.txt
