#!/usr/bin/env python3
"""Why is the first process on a fresh box often 10 % faster than the following ones (profiles/README.md, round 3)?
One process, the c3 workload re-created several times: fresh allocations each time, with and without returning the
memory to the driver, and after idle pauses.  Prints ms/step of 200 timed steps (after 400 ageing steps) per trial."""
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


def trial(tag):
    env = BatchedAntsEnv(cfg, dev)
    env.reset(init)
    for t in range(420):
        env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(200):
        env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    print("%-34s ms/step %.4f  obs@%#x ws@%#x  t=%.1fs" % (tag, ms, env.obs.data_ptr(), env._ws_ptr, time.perf_counter() - T0), flush=True)
    del env


T0 = time.perf_counter()
trial("first")
trial("second (allocator re-uses blocks)")
torch.cuda.empty_cache()
trial("after empty_cache (fresh hipMalloc)")
junk = [torch.empty(1 << 28, dtype=torch.uint8, device=dev) for _ in range(12)]  # 3 GiB of other allocations first
trial("behind 3 GiB of other allocations")
del junk
torch.cuda.empty_cache()
for pause in (10, 30):
    time.sleep(pause)
    trial("after %d s idle" % pause)
for i in range(3):
    trial("back to back %d" % i)
