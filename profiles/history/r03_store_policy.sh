#!/bin/bash
# cache-policy bits of the 16-byte copy-out stores (raw buffer stores: 1 sc0, 2 nt, 16 sc1), exact-byte plan (e_) and whole-line ablation (l_), beside the gathers
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{ echo "# c3"; VARIANTS="d2 d2_lines e_aux2 l_aux2 e_aux0 l_aux0 e_aux1 l_aux1 e_aux3 l_aux3 e_aux16 l_aux16 e_aux17 l_aux17 e_aux18 l_aux18" ROUNDS=1 bash profiles/abn.sh --config c3; 
  echo "# c3 again"; VARIANTS="d2 e_aux2 l_aux2 l_aux16 l_aux17 l_aux18 e_aux16 e_aux18" ROUNDS=1 bash profiles/abn.sh --config c3; } | tee gpurun_out/r03_store_policy.txt
