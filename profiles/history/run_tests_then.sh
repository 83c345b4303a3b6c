#!/bin/bash
# GPU parity suite first; the given command only when it is green.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -6 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
"$@"
