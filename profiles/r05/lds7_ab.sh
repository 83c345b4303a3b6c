#!/bin/bash
# c5's in-loop policy launch: 23 488 bytes of LDS per workgroup admit six workgroups per CU, 23 360 (no rock masks when there
# are no rocks) seven.  base = the tree before, lds7 = with it.
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in "--config c5" "--config c5 --no-obs" "--config c2"; do echo "== $c"; bash $R/profiles/ab.sh run base lds7 3 $c --no-explicit-sweep; done
