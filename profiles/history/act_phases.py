"""Per-workgroup split of k_act's per-ant phases from the trace stamps (ACT_ABL_TRACE): entry -> phase 0 done
(bit maps / tables staged) -> phase 1 done (mandibles, food exchange) -> phase 2 done (move, frames,
presence map) -> perception done -> rewards done.   python3 profiles/act_phases.py [E N W]"""
import os, sys, ctypes as C
os.environ["ANTSRL_ABLATE"] = str(int(os.environ.get("ANTSRL_ABLATE", "0")) | 32768)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from antsrl_amd import _lib, config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
E, N, W = (int(a) for a in (sys.argv[1:4] + ["1024", "512", "256"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
cfg = cm.make_cfg(E, N, W, W, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
env = BatchedAntsEnv(cfg, dev); env.reset(synth_init(cfg, seed=1234))
g = torch.Generator(device=dev); g.manual_seed(99)
rot = torch.randint(-1, 2, (4, E, N), generator=g, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (4, E, N), generator=g, device=dev, dtype=torch.int8)
for t in range(12): env.step_update(rot[t % 4], ph[t % 4], None)
torch.cuda.synchronize()
buf = np.zeros((max(E, 1), 8), np.uint64)
assert _lib.load().antsrl_debug_read_act_trace(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), E) == 0
t = buf.astype(np.int64)
order = [0, 6, 7, 1, 2, 3]
names = ["phase 0 (staging)", "phase 1 (mandibles, food)", "phase 2 (move, frames, presence)", "perception", "rewards"]
us = t[:, order] / 100.0
print("E=%d N=%d %dx%d: kernel span %.1f us" % (E, N, W, W, us[:, -1].max() - us[:, 0].min()))
for k in range(5):
    d = us[:, k + 1] - us[:, k]
    print("  %-34s mean %6.2f us   p90 %6.2f" % (names[k], d.mean(), np.percentile(d, 90)))
