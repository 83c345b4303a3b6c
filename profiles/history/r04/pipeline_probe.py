#!/usr/bin/env python3
"""Feasibility: the c3 batch as TWO half batches (two handles, env_id_base 0 / 512, two streams) stepped in a software
pipeline — U_A | P_A || U_B | P_B || U_A' | ... (U = k_update_move, P = k_perceive): does the latency-bound U of one half hide
under the bandwidth-bound P of the other?   python3 profiles/r04/pipeline_probe.py [ring|free|single]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init

mode = sys.argv[1] if len(sys.argv) > 1 else "ring"
E, N = 1024, 512
dev = torch.device("cuda", 0)
kw = dict(n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
g = torch.Generator(device=dev); g.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=g, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (8, E, N), generator=g, device=dev, dtype=torch.int8)
AGE, STEPS = 400, 300

def run_single():
    cfg = cm.make_cfg(E, N, 256, 256, **kw)
    env = BatchedAntsEnv(cfg, dev); env.tune_placement(); env.reset(synth_init(cfg, seed=1234))
    for t in range(AGE): env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(STEPS): env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / STEPS * 1e3, float(env.obs[E // 2, 7].sum()), float(env.reward.sum())

def run_pipe(ring):
    """ring: the library's timing hook records a caller event between U and P of every call (slot [2], include/antsrl.h): half
    B's call waits for half A's U of the same step, half A's next call for half B's U — U of one half beside P of the other."""
    H = E // 2
    envs, st = [], [torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev, priority=-1)]
    for k in range(2):
        cfg = cm.make_cfg(H, N, 256, 256, env_id_base=k * H, n_envs_total=E, **kw)
        e = BatchedAntsEnv(cfg, dev); e.tune_placement(); e.reset(synth_init(cfg, seed=1234, env_offset=k * H)); envs.append(e)
    torch.cuda.synchronize()
    sl = [slice(0, H), slice(H, E)]
    rots = [[rot[i, s].contiguous() for i in range(8)] for s in sl]
    phs = [[ph[i, s].contiguous() for i in range(8)] for s in sl]
    NEV = cm.TIMING_EVENTS
    # [half][parity][slot]: torch events whose raw handles the library records (created by a first record)
    evs = [[[torch.cuda.Event() for _ in range(NEV)] for _ in range(2)] for _ in range(2)]
    for k in range(2):
        for p in range(2):
            for e in evs[k][p]: e.record(st[k])
    torch.cuda.synchronize()
    def step(t):
        for k in range(2):
            with torch.cuda.stream(st[k]):
                if ring:
                    other = evs[1 - k][t % 2][2] if k == 1 else evs[1][(t + 1) % 2][2]  # B: A's U(t);  A: B's U(t - 1)
                    st[k].wait_event(other)
                    envs[k].set_timing_events([e.cuda_event for e in evs[k][t % 2]])
                envs[k].step_update(rots[k][t % 8], phs[k][t % 8], None)
    for t in range(AGE): step(t)
    torch.cuda.synchronize()
    off = float(os.environ.get("OFFSET_US", "0"))
    if off > 0:  # half B starts late by `off` us, once: two free-running streams of equal period keep their phase
        with torch.cuda.stream(st[1]): torch.cuda._sleep(int(off * 2400))
    t0 = time.perf_counter()
    for t in range(STEPS): step(t)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / STEPS * 1e3
    return ms, float(envs[1].obs[0, 7].sum()), float(envs[0].reward.sum() + envs[1].reward.sum())

if mode == "single": r = run_single()
else: r = run_pipe(mode == "ring")
print("%-7s ms per full step %.4f   (obs checksum %.3f, reward sum %.1f)" % (mode, r[0], r[1], r[2]))
