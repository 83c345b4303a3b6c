#!/bin/bash
# the wave time line says the whole-line copy-out makes the GATHERS slower (group 3 waits 0.9 us for its records): does a deeper gather pipeline turn it around?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{ echo "# c3"; VARIANTS="d2 d3 d2_lines d3_lines" ROUNDS=2 bash profiles/abn.sh --config c3; } | tee gpurun_out/r03_lines_depth.txt
