#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
hipcc --offload-arch=gfx950 -O3 -o /tmp/obs_stream_probe $R/profiles/obs_stream_probe.hip || exit 1
P=/tmp/obs_stream_probe
for run in 8 32; do for map in 0 1; do $P --lds 21 --run $run --map $map; done; done
for run in 2 8 32 128; do for map in 0 1; do $P --lds 21 --run $run --map $map --blockflush 1; done; done
$P --lds 21 --run 8 --map 0 --blockflush 1 --nt 0
$P --lds 0 --run 8 --map 0 --blockflush 1
$P --lds 21 --run 8 --map 0 --blockflush 1 --wpb 8
$P --lds 21 --run 4 --map 0 --wpb 8
$P --lds 21 --run 8 --map 0 --group 1
python3 - <<'PY'
import torch
x = torch.empty(1024*512*343, dtype=torch.float32, device="cuda")
for _ in range(3): x.fill_(1.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): x.fill_(2.0)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 10
print("torch fill_ %.4f ms %.2f TB/s" % (ms, x.numel() * 4 / ms / 1e9))
PY
