#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
hipcc --offload-arch=gfx950 -O3 -o /tmp/fill_clone $R/profiles/fill_clone_probe.hip || exit 1
for nt in 0 1; do for unr in 1 2 4 8; do /tmp/fill_clone --unr $unr --nt $nt; done; done
for ch in 2 4 11 44; do /tmp/fill_clone --unr 4 --chunks $ch --nt 1; /tmp/fill_clone --unr 4 --chunks $ch --nt 0; done
