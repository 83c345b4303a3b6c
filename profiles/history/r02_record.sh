#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?; tail -3 gpurun_out/gpu_tests.log; [ $rc -ne 0 ] && exit $rc
bash profiles/step_drift.sh > gpurun_out/step_drift.txt; cat gpurun_out/step_drift.txt
bash profiles/all_configs_ab.sh > gpurun_out/all_configs_ab.txt; cat gpurun_out/all_configs_ab.txt
