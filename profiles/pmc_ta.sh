#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/profiles/pmc_pass.sh ta1 TA_TA_BUSY_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum 2>&1 | grep "k_perceive\|k_move\|rror"
bash $R/profiles/pmc_pass.sh ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TD_TD_BUSY_sum 2>&1 | grep "k_perceive\|k_move\|rror"
bash $R/profiles/pmc_pass.sh ta3 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM SQ_WAVES SQ_LEVEL_WAVES TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum 2>&1 | grep "k_perceive\|k_move\|rror"
tail -3 $R/gpurun_out/pmc_ta1.log
