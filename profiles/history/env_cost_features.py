"""What makes an environment expensive for k_act?  Correlates the per-env perception time (phase
timeline, ANTSRL_ABLATE=32768) with features of the env's state: spread of the ants (gather
locality), ants whose patch can touch a rock, unexplored cells seen, ants per occupied cell."""
import os, sys, ctypes as C
os.environ["ANTSRL_ABLATE"] = "32768"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from antsrl_amd import _lib, config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
E = 1024
cfg = cm.make_cfg(E, 512, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
dev = torch.device("cuda", 0); env = BatchedAntsEnv(cfg, dev); init = synth_init(cfg, seed=1234); env.reset(init)
g = torch.Generator(device=dev); g.manual_seed(99)
rot = torch.randint(-1, 2, (4, E, cfg.n_ants), generator=g, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (4, E, cfg.n_ants), generator=g, device=dev, dtype=torch.int8)
for t in range(120): env.step_update(rot[t % 4], ph[t % 4], None)
torch.cuda.synchronize()
buf = np.zeros((E, 8), np.uint64)
assert _lib.load().antsrl_debug_read_act_trace(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), E) == 0
t = buf[:, :4].astype(np.int64)
perc = (t[:, 2] - t[:, 1]) / 100.0
early = (t[:, 0] - t[:, 0].min()) / 100.0 < 5   # first round only: same contention for all
xyt = env.read_state(cm.S_ANTS_XYT).cpu().numpy()
rocks = env.read_state(cm.S_ROCK_CENTERS).cpu().numpy(); rw = env.read_state(cm.S_ROCK_RW).cpu().numpy()
expl = env.read_state(cm.S_EXPLORED).cpu().numpy()
walls = env.read_state(cm.S_WALLS).cpu().numpy()
x, y = xyt[..., 0], xyt[..., 1]
spread = np.sqrt(x.var(1) + y.var(1))
cells = (x.astype(int) * 256 + y.astype(int))
distinct = np.array([len(np.unique(c)) for c in cells])
lines = np.array([len(np.unique(c >> 4)) for c in cells])           # distinct 128-byte pheromone lines under the ants
d = np.sqrt((x[:, :, None] - rocks[:, None, :, 0]) ** 2 + (y[:, :, None] - rocks[:, None, :, 1]) ** 2)
near_rock = (d < rw[:, None, :, 0] + 10).any(2).sum(1)
explored_frac = expl.reshape(E, -1).mean(1)
feat = dict(spread=spread, distinct_cells=distinct, distinct_lines=lines, ants_near_rock=near_rock, explored_frac=explored_frac,
            wall_frac=walls.reshape(E, -1).mean(1))
print("round-1 workgroups: %d, perception mean %.1f us, std %.1f" % (early.sum(), perc[early].mean(), perc[early].std()))
for k, v in feat.items():
    print("  corr(perception, %-16s) = %+.3f   (feature mean %.3g, std %.3g)" % (k, np.corrcoef(perc[early], v[early])[0, 1], v.mean(), v.std()))
hw = buf[:, 4]; xcc = (buf[:, 5] & 0xF).astype(int)
cu = ((hw >> 8) & 0xF).astype(int); sh = ((hw >> 12) & 1).astype(int); se = ((hw >> 13) & 7).astype(int)
cu_uid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("by XCD (round 1):", [round(float(perc[early & (xcc == k)].mean()), 1) for k in range(8)])
m = np.array([perc[early & (cu_uid == c)].mean() for c in np.unique(cu_uid[early])])
print("by CU (round 1): %d CUs, mean of CU means %.1f, std of CU means %.1f, min %.1f, max %.1f" % (len(m), m.mean(), m.std(), m.min(), m.max()))
# the two workgroups that share a CU in round 1: how similar are they?
pairs = [perc[early & (cu_uid == c)] for c in np.unique(cu_uid[early])]
pairs = np.array([p_ for p_ in pairs if len(p_) == 2])
print("within-CU pair difference: mean |d| %.1f us; correlation between the two %.3f" % (np.abs(pairs[:, 0] - pairs[:, 1]).mean(), np.corrcoef(pairs[:, 0], pairs[:, 1])[0, 1]))
print("by SE:", [round(float(perc[early & (se == k)].mean()), 1) for k in range(8) if (early & (se == k)).any()])
print("corr(perception, env index) %.3f" % np.corrcoef(perc[early], np.arange(E)[early])[0, 1])
