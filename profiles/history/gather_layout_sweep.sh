#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_layout_probe $R/profiles/gather_layout_probe.hip || exit 1
P=/tmp/gather_layout_probe
for rep in 1 2; do for spread in 128 48 16; do for l in 0 1 2 3 4 5; do $P --layout $l --spread $spread; done; done; done
for l in 0 1 5; do $P --layout $l --spread 128 --envs 512; done
