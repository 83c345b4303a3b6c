#!/bin/bash
# Texture-addresser / texture-data counters of k_perceive / k_move, TWO per pass.
# Round 2 asked for four TA_*_sum counters in one pass: rocprofv3 aborted at the first dispatch with
#   "Could not construct profile cfg failed with error code 38: Request exceeds the capabilities of the hardware to collect"
# (gpurun_out/pmc_ta1.log of that round) — the TA block has two counter slots per instance, and the tool turns the refusal
# into abort() instead of skipping the counter.  Two per pass fit.  (Pass 3 of the old script — GRBM / SQ / TCP counters —
# worked and is in profiles/r02/pmc_diag.txt.)
R=${GRAFT_REPO_ROOT:-/root/repo}
for pair in "TA_TA_BUSY_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
            "TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum"; do
  tag=ta_$(echo $pair | tr ' ' '_' | cut -c1-40)
  bash $R/profiles/pmc_pass.sh $tag $pair 2>&1 | grep "k_perceive\|k_move\|k_update\|rror" || { echo "pass '$pair' failed: see gpurun_out/pmc_$tag.log"; exit 1; }
done
