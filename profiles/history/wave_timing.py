"""Per-wave time split of k_act's perception loop (instrumented build, not in the tree):

    git apply profiles/wave_timing.patch && profiles/ab.sh build timing && git checkout antsrl_amd/csrc/antsrl_act.hip
    gpurun -- 'ANTSRL_LIB=$GRAFT_REPO_ROOT/antsrl_amd/lib/variants/timing.so python3 profiles/wave_timing.py'

The patch brackets the four parts of an iteration with s_memrealtime (100 MHz) and accumulates them per
wave: issuing the next group's gathers (address math + 4 loads), waiting for the current group's gather
data, channel values -> LDS staging, and the flush (3 LDS reads + 3 streaming stores)."""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from antsrl_amd import _lib, config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
E=1024
cfg = cm.make_cfg(E, 512, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
dev=torch.device("cuda",0); env = BatchedAntsEnv(cfg, dev); env.reset(synth_init(cfg, seed=1234))
g = torch.Generator(device=dev); g.manual_seed(99)
rot = torch.randint(-1, 2, (4, E, cfg.n_ants), generator=g, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (4, E, cfg.n_ants), generator=g, device=dev, dtype=torch.int8)
for t in range(40): env.step_update(rot[t % 4], ph[t % 4], None)
torch.cuda.synchronize()
buf = np.zeros((8192, 8), np.uint64)
assert _lib.load().antsrl_debug_read_act_trace(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 8192) == 0
t = buf[:E*8, :5].astype(np.float64) / 100.0  # us
print("per wave (us, mean over %d waves): fetch-issue %.1f  wait-for-gathers %.1f  compute+staging %.1f  flush+stores %.1f  loop total %.1f" % ((len(t),) + tuple(t.mean(0))))
print("p90:", np.percentile(t, 90, axis=0).round(1))
