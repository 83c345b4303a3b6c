#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for c in c3 c2 c5 c5a; do timeout -k 10 200 python3 profiles/prc_trace.py $c 2>&1 | grep -v Warning; done | tee gpurun_out/r03_prc_trace.txt
