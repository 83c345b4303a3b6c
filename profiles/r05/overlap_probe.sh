#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export ANTSRL_LIB=$R/antsrl_amd/lib/variants/umlds.so
for i in 1 2; do
  python3 $R/profiles/r05/overlap_probe.py single
  ANTSRL_PRC_LDS_PAD=15 python3 $R/profiles/r05/overlap_probe.py single
  python3 $R/profiles/r05/overlap_probe.py free
  ANTSRL_PRC_LDS_PAD=15 python3 $R/profiles/r05/overlap_probe.py free
  ANTSRL_PRC_LDS_PAD=20 python3 $R/profiles/r05/overlap_probe.py free
done
