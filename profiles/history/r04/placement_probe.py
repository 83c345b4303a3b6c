#!/usr/bin/env python3
"""Which buffer's placement makes c3 processes differ by +-10 %?  ("where the buffers lie in device memory".)
ONE process, one deterministic workload (same reset, same actions, same age): only WHERE the workspace and the
observation tensor lie changes between trials.
  phase 1  one workspace; the observation tensor re-pointed inside ONE big buffer at different byte offsets
  phase 2  one workspace; the observation tensor in freshly allocated buffers (other allocations made in between)
  phase 3  fresh workspaces (other allocations made in between), the observation tensor fixed
Per trial: mean k_perceive / k_update_move (HIP events) over 40 steps after the same 300-step ageing.
    python profiles/r04/placement_probe.py [--envs 1024]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
from bench import HipEvents

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=1024)
ap.add_argument("--age", type=int, default=300)
ap.add_argument("--steps", type=int, default=40)
args = ap.parse_args()
E, N = args.envs, 512
cfg = cm.make_cfg(E, N, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
init = synth_init(cfg, seed=1234)
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (8, E, N), generator=gen, device=dev, dtype=torch.int8)
NEV = cm.TIMING_EVENTS
evs = HipEvents(NEV * (args.steps // 5 + 1))
row = 343
nobs = E * N * row


def run(env):
    env.reset(init)
    for t in range(args.age):
        env.step_update(rot[t % 8], ph[t % 8], None)
    slots = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.steps):
        if t % 5 == 0:
            env.set_timing_events([evs.ev[NEV * len(slots) + i].value for i in range(NEV)])
            slots.append(len(slots))
        env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    kp = float(np.mean([evs.elapsed_ms(NEV * j + 2, NEV * j + 3) for j in slots]))
    ku = float(np.mean([evs.elapsed_ms(NEV * j + 1, NEV * j + 2) for j in slots]))
    return ms, kp, ku


env = BatchedAntsEnv(cfg, dev)
print("workspace VA %x (%d MiB), obs VA %x" % (env._ws_ptr, env.workspace_bytes >> 20, env.obs.data_ptr()))
print("baseline (own obs):            %.4f ms/step  k_perceive %.4f  k_update_move %.4f" % run(env))
print("baseline again:                %.4f ms/step  k_perceive %.4f  k_update_move %.4f" % run(env))
own = env.obs
print("# phase 1: one big buffer, the observation tensor at different offsets")
big = torch.empty(nobs + (64 << 20), dtype=torch.float32, device=dev)
for off_b in (0, 4096, 65536, 1 << 20, (2 << 20) + 4096, 16 << 20, (64 << 20) - 128, 128, 2048):
    o = off_b // 4
    env.obs = big[o:o + nobs].view(own.shape)
    print("  offset %9d B  VA %x:  %.4f ms/step  k_perceive %.4f  k_update_move %.4f" % ((off_b, env.obs.data_ptr()) + run(env)))
del big
print("# phase 2: fresh observation buffers")
keep = []
for i in range(6):
    keep.append(torch.empty(int(np.random.default_rng(i).integers(1, 900)) << 20, dtype=torch.uint8, device=dev))  # shifts what comes next
    b = torch.empty(nobs, dtype=torch.float32, device=dev)
    env.obs = b.view(own.shape)
    print("  obs VA %x:  %.4f ms/step  k_perceive %.4f  k_update_move %.4f" % ((env.obs.data_ptr(),) + run(env)))
    keep.append(b)
env.obs = own
print("# phase 3: fresh workspaces (obs = the first env's own tensor)")
for i in range(6):
    keep.append(torch.empty(int(np.random.default_rng(100 + i).integers(1, 900)) << 20, dtype=torch.uint8, device=dev))
    e2 = BatchedAntsEnv(cfg, dev)
    e2.obs = own
    print("  workspace VA %x:  %.4f ms/step  k_perceive %.4f  k_update_move %.4f" % ((e2._ws_ptr,) + run(e2)))
    keep.append(e2)
print("first env again:               %.4f ms/step  k_perceive %.4f  k_update_move %.4f" % run(env))
