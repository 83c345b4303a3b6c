#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/r03_tests_final.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_tests_final.log
CONFIGS="c3 c2 c5 c4" TAG=final bash profiles/r03_final_others.sh > /dev/null 2>&1
bash profiles/r03_all_configs.sh 2>&1 | tee gpurun_out/r03_all_configs.txt | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
