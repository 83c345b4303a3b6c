#!/bin/bash
# where do k_update_move's waves spend their time?  (two PMC passes; the third — TA — two counters only)
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/profiles/pmc_pass.sh um_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD | grep "k_update_move\|k_perceive"
bash $R/profiles/pmc_pass.sh um_lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA | grep "k_update_move\|k_perceive"
bash $R/profiles/pmc_pass.sh um_tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum | grep "k_update_move\|k_perceive"
