#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
hipcc --offload-arch=gfx950 -O3 -o /tmp/obs_stream_probe $R/profiles/obs_stream_probe.hip || exit 1
P=/tmp/obs_stream_probe
$P --lds 21
$P --lds 21 --dup 1
$P --lds 21 --dup 2
$P --lds 21 --dup 1 --work 8
$P --lds 21 --dup 2 --work 8
$P --lds 21 --dup 1 --nt 0
$P --lds 21 --dup 1 --gather 2 --work 8
