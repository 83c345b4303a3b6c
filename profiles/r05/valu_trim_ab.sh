#!/bin/bash
# k_perceive with fewer vector instructions per cell (power-of-two grids: the slot's quarter-rate multiply as a shift; the
# wall / area bits as extract + convert; the explored test without its shift): base = the tree before, valu = with them.
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in "--config c5" "--config c5 --no-obs" "--config c2" "--config c3" "--config c4"; do echo "== $c"; bash $R/profiles/ab.sh run base valu 2 $c --no-explicit-sweep; done
