"""c3 stepped as S independent handles of 1024/S envs on S HIP streams (ping-pong vectorised envs): does
overlapping one part's latency-bound phases (k_update_one, k_act's per-ant phases, kernel tails) with another
part's HBM-bound perception pay?   gpurun -- 'python3 profiles/two_streams.py'"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init

dev = torch.device("cuda", 0)
E, N, K, WU = 1024, 512, 200, 20

def run(S, order="step"):
    Es = E // S
    cfg = cm.make_cfg(Es, N, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
    envs, streams, acts = [], [], []
    for s in range(S):
        env = BatchedAntsEnv(cfg, dev)
        env.reset(synth_init(cfg, seed=1234, env_offset=s * Es))
        g = torch.Generator(device=dev); g.manual_seed(99 + s)
        acts.append((torch.randint(-1, 2, (8, Es, N), generator=g, device=dev, dtype=torch.int8),
                     torch.randint(0, 3, (8, Es, N), generator=g, device=dev, dtype=torch.int8)))
        envs.append(env); streams.append(torch.cuda.Stream(dev))
    torch.cuda.synchronize()
    main = torch.cuda.current_stream(dev)
    def step(t):
        if order == "join":  # fork from / join into ONE caller stream every step: no overlap across steps
            ev = torch.cuda.Event(); ev.record(main)
        for s in range(S):
            with torch.cuda.stream(streams[s]):
                if order == "join":
                    streams[s].wait_event(ev)
                envs[s].step_update(acts[s][0][t % 8], acts[s][1][t % 8], None)
                if order == "join":
                    e2 = torch.cuda.Event(); e2.record(streams[s]); main.wait_event(e2)
    for t in range(WU): step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(K): step(t)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3

for rep in range(2):
    for S, order in ((1, "step"), (4, "step"), (4, "join"), (2, "join"), (8, "step")):
        ms = run(S, order)
        print("S=%d handles x %d envs%s: %.4f ms per step of all 1024 envs  (%.3g ant-steps/s)" % (
            S, E // S, " (fork/join on one stream every step)" if order == "join" else "", ms, E * N / ms * 1e3))
