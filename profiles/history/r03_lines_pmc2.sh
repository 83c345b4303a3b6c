#!/bin/bash
# L2 (TCC) stall / latency / queue-level counters of k_perceive: the product's exact-byte copy-out (d2) against the whole-line ablation (d2_lines)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in d2 d2_lines; do
  export ANTSRL_LIB=$R/antsrl_amd/lib/variants/$v.so
  echo "== $v"
  bash profiles/pmc_pass.sh ${v}_ea TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum | grep k_perceive
  bash profiles/pmc_pass.sh ${v}_st TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum | grep k_perceive
  bash profiles/pmc_pass.sh ${v}_lat TCC_READ_REQ_LATENCY_sum TCC_WRITE_REQ_LATENCY_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum | grep k_perceive
  bash profiles/pmc_pass.sh ${v}_cnt TCC_CYCLE_sum TCC_BUSY_sum TCC_READ_REQ_sum TCC_WRITE_REQ_sum | grep k_perceive
done 2>&1 | tee gpurun_out/r03_lines_pmc2.txt
