"""Would processing an env's ants in spatial order pay?  c3 with the ants of every env sorted by position at
reset (index order = spatial order; they move <= 1 cell per step, so it holds for the run) against the usual
random order, for the streaming (nt) and the cached observation stores (library variants built by ab.sh).
   ANTSRL_LIB=.../variants/<nt|cached>.so python3 profiles/sorted_ants_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
dev = torch.device("cuda", 0)
E, N = 1024, 512
cfg = cm.make_cfg(E, N, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
base = synth_init(cfg, seed=1234)
g = torch.Generator(device=dev); g.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=g, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (8, E, N), generator=g, device=dev, dtype=torch.int8)
def run(order):
    init = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in base.items()}
    if order != "random":
        xyt = init["ants_xyt"]
        tx, ty = (xyt[..., 0].astype(int) >> 2), (xyt[..., 1].astype(int) >> 3)   # 4 x 8 cell tiles, row-major
        key = tx * 64 + ty
        idx = np.argsort(key, axis=1, kind="stable")
        init["ants_xyt"] = np.take_along_axis(xyt, idx[..., None], axis=1)
        init["seed"] = np.take_along_axis(init["seed"], idx, axis=1)
    env = BatchedAntsEnv(cfg, dev); env.reset(init)
    for t in range(10): env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(100): env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 100 * 1e3
for rep in range(2):
    for order in ("random", "sorted"):
        print("%s  ants in %-6s order: %.4f ms per step" % (os.path.basename(os.environ.get("ANTSRL_LIB", "default")), order, run(order)))
