#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
L=${1:-$R/antsrl_amd/lib/libantsrl_hip.so}
export ANTSRL_LIB=$L
for i in 1 2; do
  python3 $R/profiles/r05/overlap_probe.py single 2>/dev/null
  python3 $R/profiles/r05/overlap_probe.py free 2 2>/dev/null
  python3 $R/profiles/r05/overlap_probe.py free 4 2>/dev/null
  python3 $R/profiles/r05/overlap_probe.py free 8 2>/dev/null
done
