#!/bin/bash
# On a box where BatchedAntsEnv's default placement (workspace torch.empty, outputs antsrl_mem_alloc) is SLOW, which
# allocation is fast, and how reliably?  Arms of profiles/r04/placement_probe4.py in fresh processes (every arm is a new
# draw of physical pages); on a fast box only the short list runs.
#   bash profiles/r04/placement_probe5.sh > gpurun_out/r04_placement_probe5_$(date +%H%M).txt   (gpurun_out/ is per call: one file per box)
R=${GRAFT_REPO_ROOT:-/root/repo}
P="python3 $R/profiles/r04/placement_probe4.py"
first=$($P product 2>/dev/null | tail -n 1)
echo "== $(date +%H:%M:%S) $first"
python3 $R/profiles/hbm_bw_probe.py 2>/dev/null | tr '\n' ';'; echo
rocm-smi --showclocks 2>/dev/null | grep -i "mclk\|sclk\|fclk" | head -4
rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -i "partition" | head -4
rocm-smi --showmeminfo vram 2>&1 | grep -i "vram" | head -3
rocm-smi --showtemp --showpower 2>&1 | grep -i "temp\|power" | head -8
kp=$(echo "$first" | sed -n 's/.*k_perceive \([0-9.]*\).*/\1/p')
if python3 -c "import sys; sys.exit(0 if float('${kp:-0}') > 0.185 else 1)"; then
  echo "   slow box"
  for i in 1 2 3 4; do CHUNK_MB=2 $P vmm_seq 2>/dev/null | tail -n 1; done
  for i in 1 2 3; do SEED=$i CHUNK_MB=2 $P vmm_shuf 2>/dev/null | tail -n 1; done
  for i in 1 2 3; do CHUNK_MB=4 $P vmm_seq 2>/dev/null | tail -n 1; done
  for i in 1 2; do CHUNK_MB=8 $P vmm_seq 2>/dev/null | tail -n 1; done
  for i in 1 2; do $P product 2>/dev/null | tail -n 1; done
else
  echo "   fast box"
  for i in 1 2 3; do CHUNK_MB=2 $P vmm_seq 2>/dev/null | tail -n 1; done
  $P own 2>/dev/null | tail -n 1
fi
