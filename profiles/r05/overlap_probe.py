#!/usr/bin/env python3
"""k_update_move of one half batch UNDER k_perceive of the other, with room made for it: the c3 batch as two handles of 512
environments (env_id_base 0 / 512) on two free-running streams — round 4's pipeline_probe (−3 %: k_update_move's 512-thread
workgroups found no room while k_perceive held seven workgroups per CU) — with k_perceive capped at FIVE workgroups per CU by an
LDS pad (ANTSRL_PRC_LDS_PAD=15, profiling build) so that one k_update_move workgroup (8 waves, 2 per SIMD, 128 of the SIMD's 512
vector registers, 17 KB of LDS without the frame parking) always fits beside them.
    ANTSRL_LIB=antsrl_amd/lib/variants/umlds.so [ANTSRL_PRC_LDS_PAD=15] python3 profiles/r05/overlap_probe.py single|free"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init

mode = sys.argv[1] if len(sys.argv) > 1 else "free"
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
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3, float(env.obs[E // 2, 7].sum()), float(env.reward.sum())


def run_free():
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 2  # handles / streams
    H = E // S
    envs, st = [], [torch.cuda.Stream(device=dev, priority=-1) for _ in range(S)]
    for k in range(S):
        cfg = cm.make_cfg(H, N, 256, 256, env_id_base=k * H, n_envs_total=E, **kw)
        e = BatchedAntsEnv(cfg, dev); e.tune_placement(); e.reset(synth_init(cfg, seed=1234, env_offset=k * H)); envs.append(e)
    torch.cuda.synchronize()
    sl = [slice(k * H, (k + 1) * H) for k in range(S)]
    rots = [[rot[i, s].contiguous() for i in range(8)] for s in sl]
    phs = [[ph[i, s].contiguous() for i in range(8)] for s in sl]

    def step(t):
        for k in range(S):
            with torch.cuda.stream(st[k]):
                envs[k].step_update(rots[k][t % 8], phs[k][t % 8], None)
    for t in range(AGE): step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(STEPS): step(t)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3, float(envs[1].obs[0, 7].sum()), float(sum(e.reward.sum() for e in envs))


r = run_single() if mode == "single" else run_free()
print("%-7s %s pad %-3s ms per full step %.4f   (obs checksum %.3f, reward sum %.1f)" % (mode, (sys.argv[2] if len(sys.argv) > 2 else ""), os.environ.get("ANTSRL_PRC_LDS_PAD", "0"), r[0], r[1], r[2]), flush=True)
