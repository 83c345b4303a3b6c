#!/bin/bash
# The GPU suite once per alternate code path (each switch is read once per process):
#   fused update at the tail of k_act, the per-phase-loop update kernel for every N, the chunked policy
#   kernel for every feature count, the general stencil for rank-1 filters, the float64 LDS-tiled sweep, separate food / pheromone arrays.
# Usage (on a GPU box):  bash tests/alt_paths.sh
cd "$(dirname "$0")/.."
rc=0
for v in ANTSRL_FUSE_UPDATE ANTSRL_UPDATE_LOOPS ANTSRL_POLICY_CHUNKED ANTSRL_NO_SEPARABLE ANTSRL_SWEEP_TILED ANTSRL_NO_INTERLEAVE; do
  echo "== $v=1"
  env $v=1 python3 -m pytest tests -m gpu -q 2>&1 | tail -2 || rc=1
done
exit $rc
