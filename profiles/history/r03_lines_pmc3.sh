#!/bin/bash
# vector L1 (TCP) / texture-addresser counters of k_perceive: exact-byte copy-out (d2) against the whole-line ablation (d2_lines)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in d2 d2_lines; do
  export ANTSRL_LIB=$R/antsrl_amd/lib/variants/$v.so
  echo "== $v"
  bash profiles/pmc_pass.sh ${v}_tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum | grep k_perceive
  bash profiles/pmc_pass.sh ${v}_tcp2 TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_NC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum | grep k_perceive
  bash profiles/pmc_pass.sh ${v}_tcp3 TCP_TCC_WRITE_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum | grep k_perceive
  bash profiles/pmc_pass.sh ${v}_sq SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_ANY | grep k_perceive
done 2>&1 | tee gpurun_out/r03_lines_pmc3.txt
