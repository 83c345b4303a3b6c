#!/usr/bin/env python3
"""What a caller pays who wants the step's outputs in HOST memory (the reference's own return type: numpy arrays): the c3
step followed by the copy of observation / agent_state / reward / done into pinned host buffers, per step, measured — DESIGN.md
§1 quotes it beside the headline, never as `value`.      python3 profiles/r05/pcie_inclusive.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init

E, N = 1024, 512
dev = torch.device("cuda", 0)
cfg = cm.make_cfg(E, N, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
env = BatchedAntsEnv(cfg, dev)
env.tune_placement()
env.reset(synth_init(cfg, seed=1234))
g = torch.Generator(device=dev); g.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=g, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (8, E, N), generator=g, device=dev, dtype=torch.int8)
outs = [env.obs, env.agent_state, env.reward, env.done]
host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in outs]
nbytes = sum(t.numel() * t.element_size() for t in outs)
AGE, STEPS = 400, 40


def run(copy):
    for t in range(20):
        env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(STEPS):
        env.step_update(rot[t % 8], ph[t % 8], None)
        if copy:
            for h, d in zip(host, outs):
                h.copy_(d, non_blocking=True)
            torch.cuda.synchronize()  # (the caller reads the arrays before it chooses the next actions)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


for t in range(AGE):
    env.step_update(rot[t % 8], ph[t % 8], None)
dev_ms = run(False)
host_ms = run(True)
print("c3 (1024 envs x 512 ants): step on the device %.4f ms = %.3g ant-steps/s" % (dev_ms, E * N / dev_ms * 1e3))
print("  + outputs into pinned host memory (%.1f MB per step): %.3f ms per step = %.3g ant-steps/s; the copy alone moves %.1f GB/s" % (
    nbytes / 1e6, host_ms, E * N / host_ms * 1e3, nbytes / ((host_ms - dev_ms) * 1e-3) / 1e9))
