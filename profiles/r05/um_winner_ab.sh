#!/bin/bash
# k_update_move: the last-writer-wins verdict forwarded from the update to the move (no second hash table) — A/B of two
# library builds (um_old = the tree before, um_new = with it), alternating on one device.  profiles/ab.sh builds them.
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in c3 c2 c4 c5; do
  echo "== $c"; bash $R/profiles/ab.sh run um_old um_new 2 --config $c --no-explicit-sweep
done
