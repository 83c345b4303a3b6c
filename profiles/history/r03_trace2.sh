#!/bin/bash
# the wave time line of the product's exact-byte copy-out against the whole-line ablation (every line written once, wrong bytes): which phase pays?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for v in trace trace_lines trace trace_lines; do echo "== $v"; TRACE_VARIANT=$v timeout -k 10 200 python3 profiles/prc_trace.py c3 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee gpurun_out/r03_prc_trace_lines.txt
