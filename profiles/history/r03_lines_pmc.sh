#!/bin/bash
# why is the whole-line copy-out (PRC_FLUSH_LINES) slower inside k_perceive although its store stream alone is faster?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in base_r3 lines2; do
  export ANTSRL_LIB=$R/antsrl_amd/lib/variants/$v.so
  echo "== $v"
  bash profiles/pmc_pass.sh ${v}_f FETCH_SIZE | grep "k_perceive"; bash profiles/pmc_pass.sh ${v}_w WRITE_SIZE | grep "k_perceive"
  bash profiles/pmc_pass.sh ${v}_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum | grep "k_perceive"
  bash profiles/pmc_pass.sh ${v}_sq SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD | grep "k_perceive"
done
