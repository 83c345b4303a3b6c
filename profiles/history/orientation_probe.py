"""Does k_act's time depend on how the 7x7 patch lies against the memory rows?  c3 with every ant at the same
fixed heading (no rotation actions), timed per heading.   gpurun -- 'python3 profiles/orientation_probe.py'"""
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
rot = torch.zeros((E, N), dtype=torch.int8, device=dev)
ph = torch.ones((E, N), dtype=torch.int8, device=dev)
for rep in range(2):
    for deg in (0, 15, 30, 45, 60, 90, "random"):
        init = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in base.items()}
        if deg != "random":
            init["ants_xyt"][..., 2] = np.deg2rad(deg)
        env = BatchedAntsEnv(cfg, dev); env.reset(init)
        for t in range(5): env.step_update(rot, ph, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(40): env.step_update(rot, ph, None)
        torch.cuda.synchronize()
        print("heading %-6s: %.4f ms per step" % (deg, (time.perf_counter() - t0) / 40 * 1e3))
        del env
