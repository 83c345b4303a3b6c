#!/bin/bash
# The GPU suite once per alternate code path.  The switches exist only in the PROFILING library
# (libantsrl_hip_prof.so, `python -m antsrl_amd.build --prof`; the product library reads no environment variable);
# each is read once per process:
#   fused update at the tail of k_act, the per-phase-loop update kernel for every N, the chunked policy kernel for every
#   feature count, the general stencil for rank-1 filters, the float64 LDS-tiled sweep, separate food / pheromone arrays,
#   the one-column march instead of the two-column stencils, 7-row march segments (every segment edge), 32 ants per wave in
#   k_perceive, every update enqueued at once (no k_update_move); round 3: row-major cell records (no 2 x 4-cell blocks), four
#   stacked segments per workgroup in the separable stencil, its minimal column halo, a run of 12 ants per wave (two chunks of the
#   gather pipeline, the second one partial), a run of 5 (odd runs; the prologue wave's slot -> (wave, ant) mapping with a divisor that is
#   not a power of two); round 4: the perception frames built by k_update_move for EVERY deferred step (the product: only with
#   an in-loop policy), and never.
# (The two act paths — k_move + k_perceive / k_act — need no switch: the fixture tests pin both through AntsCfg.act_path.)
# Usage (on a GPU box):  bash tests/alt_paths.sh
cd "$(dirname "$0")/.."
python3 -m antsrl_amd.build --prof > /dev/null || exit 1
export ANTSRL_LIB=$PWD/antsrl_amd/lib/libantsrl_hip_prof.so
rc=0
for v in ${ALT_PATHS:-ANTSRL_FUSE_UPDATE=1 ANTSRL_UPDATE_LOOPS=1 ANTSRL_POLICY_CHUNKED=1 ANTSRL_NO_SEPARABLE=1 ANTSRL_SWEEP_TILED=1 ANTSRL_NO_INTERLEAVE=1 \
         ANTSRL_SWEEP_ONE_COLUMN=1 ANTSRL_SWEEP_SEG=7 ANTSRL_PRC_RUN=32 ANTSRL_NO_DEFER_UPDATE=1 \
         ANTSRL_NO_TILED=1 ANTSRL_SWEEP_STACK=1 ANTSRL_SWEEP_MINHALO=1 ANTSRL_PRC_RUN=12 ANTSRL_PRC_RUN=5 ANTSRL_FRAMES=1 ANTSRL_FRAMES=0}; do
  echo "== $v"
  env $v python3 -m pytest tests -m gpu -q -x --deselect tests/test_a_gpu_dist.py 2>&1 | grep -v '^\.\|^$' | tail -4
  [ ${PIPESTATUS[0]} -ne 0 ] && rc=1
done
exit $rc
