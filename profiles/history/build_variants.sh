#!/bin/bash
# Builds the product + profiling libraries and the compile-time ablation variants used by profiles/prc_explore.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m antsrl_amd.build --prof 2>&1 | grep -i "error" 
python -m antsrl_amd.build --variant nostore -DPRC_ABL_NO_STORE 2>&1 | grep -i "error"
python -m antsrl_amd.build --variant nomark -DPRC_ABL_NO_MARK 2>&1 | grep -i "error"
python -m antsrl_amd.build --variant nostore_nomark -DPRC_ABL_NO_STORE -DPRC_ABL_NO_MARK 2>&1 | grep -i "error"
ls -la antsrl_amd/lib/*.so antsrl_amd/lib/variants/*.so
