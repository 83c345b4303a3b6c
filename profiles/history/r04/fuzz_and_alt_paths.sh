#!/bin/bash
# Round 4's long checks on the tree of the moment: a fuzz campaign of the library-jitter test (the deferred update, tiled
# records, ftile, random env_id_base, padded rows) + the general one, then the suite once per alternate code path.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
ANTSRL_FUZZ_BASE=${FUZZ_BASE:-20000} ANTSRL_FUZZ_CASES=${FUZZ_CASES:-6000} timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "library_jitter or vs_oracle" -p no:cacheprovider > gpurun_out/r04_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -n 3 gpurun_out/r04_fuzz.log
if [ -z "$NO_ALT" ]; then bash tests/alt_paths.sh > gpurun_out/r04_alt_paths.log 2>&1; echo "alt rc=$?"; tail -n 40 gpurun_out/r04_alt_paths.log; fi
