#!/usr/bin/env python3
"""Follow-up to box_state_probe.py on a box whose first process is already in the slow state: how much device memory has to be
skipped (allocated and held while the batch's buffers are placed) to reach memory that gives the fast state?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from antsrl_amd import config as cm  # noqa: E402
from antsrl_amd.batched import BatchedAntsEnv  # noqa: E402
from antsrl_amd.synth import synth_init  # noqa: E402

dev = torch.device("cuda", 0)
E, N = 1024, 512
cfg = cm.make_cfg(E, N, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
init = synth_init(cfg, seed=1234)
g = torch.Generator(device=dev)
g.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=g, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (8, E, N), generator=g, device=dev, dtype=torch.int8)
T0 = time.perf_counter()


def trial(tag, pad_gib):
    t_a = time.perf_counter()
    torch.cuda.empty_cache()
    pad = [torch.empty(1 << 28, dtype=torch.uint8, device=dev) for _ in range(int(pad_gib * 4))]  # held while the batch is placed
    env = BatchedAntsEnv(cfg, dev)
    del pad
    torch.cuda.empty_cache()
    t_alloc = time.perf_counter() - t_a
    env.reset(init)
    for t in range(420):
        env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(200):
        env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    print("%-24s ms/step %.4f  alloc %.2fs  obs@%#x  t=%.1fs" % (tag, ms, t_alloc, env.obs.data_ptr(), time.perf_counter() - T0), flush=True)
    del env
    torch.cuda.empty_cache()


trial("no pad", 0)
for gib in (2, 8, 32, 64, 128, 200, 0, 250):
    trial("pad %d GiB" % gib, gib)
