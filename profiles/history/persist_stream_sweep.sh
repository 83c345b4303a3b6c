#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
hipcc --offload-arch=gfx950 -O3 -o /tmp/persist_stream_probe $R/profiles/persist_stream_probe.hip || exit 1
P=/tmp/persist_stream_probe
for rep in 1 2; do
for g in 0 1; do for w in 0 8; do
  $P --persist 0 --run 8 --gather $g --work $w
  $P --persist 0 --run 2 --gather $g --work $w
  for k in 4 6 8; do $P --persist 1 --wgcu $k --gather $g --work $w; $P --persist 2 --wgcu $k --gather $g --work $w; done
done; done
done
