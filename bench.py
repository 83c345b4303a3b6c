#!/usr/bin/env python3
"""bench.py — ant-steps/s of the AntsRL environment step loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c1|c2|c4|c5]

One "step" = RLApi.step + Environment.update (main.py:98 + main.py:131 of the reference) over one
batch of E environments, through the C-ABI (antsrl_step_update).  Default workload is BASELINE.json
configs[2] — the configuration the metric is quoted on: 1024 envs x 512 ants, 256x256 grid, 2
pheromone channels, walls + food + 8 circle obstacles, ExplorationReward, uniform random policy
(actions pre-generated on the device, resident in HBM before the timed region).

For N > 1 the driver launches this file under torch.distributed.run, one rank per GPU.  The
environments are independent, so every rank steps its own E envs (weak scaling, no data-path
collective) and the only exchange is the RCCL all-gather of reward/done each step.

Prints ONE JSON line on rank 0 (see DESIGN.md for the roofline byte accounting).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (E per GPU, N, W, H, R, radius-3 filter?)
    "c1": dict(E=1, N=32, W=64, H=64, R=0, radius3=False,
               desc="BASELINE configs[0]: the reference's own case, 1 env x 32 ants, 64x64 (step latency)"),
    "c2": dict(E=256, N=256, W=256, H=256, R=0, radius3=False,
               desc="BASELINE configs[1]: 256 envs x 256 ants, 256x256, 2 pheromone channels"),
    "c3": dict(E=1024, N=512, W=256, H=256, R=8, radius3=False,
               desc="BASELINE configs[2]: 1024 envs x 512 ants, 256x256, circle_obstacles+walls+food"),
    "c4": dict(E=1024, N=1024, W=512, H=512, R=0, radius3=True,
               desc="BASELINE configs[3] per-GPU shard: 1024 envs x 1024 ants, 512x512, diffuse radius 3"),
    "c5": dict(E=512, N=512, W=256, H=256, R=0, radius3=False, policy="mlp",
               desc="BASELINE configs[4] per-GPU shard: 512 envs x 512 ants, 256x256, DQN inference in-loop (bf16)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def measured_copy_gbs(dev, nbytes=1 << 30, reps=10):
    """Device-to-device copy bandwidth on this box, GB/s of read + written bytes."""
    import torch
    src = torch.empty((nbytes // 4,), dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    e1.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def algorithmic_bytes(N, W, H, Cn, K, P=49, obs_bytes=4):
    """SURVEY.md §8(d) per env-step figure, split by the kernel that moves each term."""
    sweep = 2 * Cn * W * H * 4 + W * H                 # pheromone read+write sweep, wall mask
    act = N * P * K * obs_bytes + N * 60 + N * 12 + N * 8  # obs write, ant state r+w, agent_state+reward, food RMW
    update = N * Cn * 8                                # deposit RMW
    return dict(sweep=sweep, act=act, update=update, total=sweep + act + update)


class HipEvents:
    """Raw hipEvent_t handles (the ABI hook records them on the launch stream)."""

    def __init__(self, n):
        from antsrl_amd._lib import hip_runtime
        self.hip = hip_runtime()  # the runtime torch already loaded, not a second copy
        self.ev = []
        for _ in range(n):
            e = C.c_void_p()
            rc = self.hip.hipEventCreate(C.byref(e))
            assert rc == 0, "hipEventCreate failed: %d" % rc
            self.ev.append(e)

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        rc = self.hip.hipEventElapsedTime(C.byref(ms), self.ev[a], self.ev[b])
        assert rc == 0, "hipEventElapsedTime failed: %d" % rc
        return ms.value

    def destroy(self):
        for e in self.ev:
            self.hip.hipEventDestroy(e)


def usable_cores():
    """CPU share this process may really use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(cfg_kw, make_cfg, synth_init, random_actions, budget_s=15.0):
    """The CPU oracle (a plain-C port of the reference algorithm, pinned to the reference by the
    golden fixtures) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle.oracle import Oracle, max_threads
    cores = min(max_threads(), usable_cores())
    E = max(cores, 16)
    cfg = make_cfg(E, cfg_kw["N"], cfg_kw["W"], cfg_kw["H"], **cfg_kw["extra"])
    init = synth_init(cfg, seed=4321)
    orc = Oracle(cfg, init, n_threads=cores)
    rot, ph = random_actions(cfg, 4, seed=7)

    def one(t):
        orc.step(rot[t % 4], ph[t % 4])
        orc.update(None)

    one(0)
    t0 = time.perf_counter()
    one(1)
    one(2)
    per = (time.perf_counter() - t0) / 2
    steps = int(max(3, min(50000, budget_s / max(per, 1e-6))))
    t0 = time.perf_counter()
    for t in range(steps):
        one(t)
    dt = time.perf_counter() - t0
    return dict(value=E * cfg.n_ants * steps / dt, unit="ant-steps/s", cores=cores, kind="port",
                sample="%d envs x %d ants, %dx%d grid, %d full steps (RLApi.step + Environment.update) "
                       "of the same workload; oracle/antsrl_oracle.c, OpenMP over envs, %.1f s"
                       % (E, cfg.n_ants, cfg.w, cfg.h, steps, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--envs", type=int, default=0, help="override envs per GPU")
    ap.add_argument("--rocks", type=int, default=-1, help="override the number of circle obstacles (profiling)")
    ap.add_argument("--diffuse", type=float, default=0.0,
                    help="the reference's DIFFUSE_FACTOR (pheromone.py:5-10): 3x3 filter with this weight on the 8 neighbours")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--policy", default=None, choices=["random", "mlp"],
                    help="mlp: the reference's linear DQN net evaluated in-loop on the GPU (bf16 MFMA)")
    ap.add_argument("--obs-dtype", default=None, choices=["f32", "bf16"],
                    help="observation tensor format (default: f32; c5, whose bf16 policy rounds its input anyway: bf16)")
    ap.add_argument("--explicit-sweep", action="store_true",
                    help="force the per-step pheromone sweep kernel (default: scaled units, no sweep)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # ANTSRL_BENCH_FORCE_DIST=1: run the N > 1 code path (RCCL process group, reward/done all-gather,
    # max-over-ranks timing) with a single rank — a rehearsal of the multi-GPU launch on a one-GPU box
    force_dist = world == 1 and os.environ.get("ANTSRL_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)  # backend "nccl" is RCCL on ROCm

    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init

    W_ = dict(CONFIGS[args.config])
    if args.rocks >= 0:
        W_["R"] = args.rocks
    E = args.envs or W_["E"]
    extra = dict(n_rocks=W_["R"], deposit_strength=256.0, max_time=1 << 30,
                 phero_mode=cm.PHERO_EXPLICIT_SWEEP if args.explicit_sweep else cm.PHERO_AUTO)
    if W_["radius3"]:
        ax = np.arange(-3, 4)
        g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 4.5)
        extra["filt"] = g / g.sum() * (1 - 0.001)
    if args.diffuse > 0:
        f3 = np.ones((3, 3)) * args.diffuse
        f3[1, 1] = 1 - 8 * args.diffuse
        extra["filt"] = f3 * (1 - 0.001)  # DIFFUSE_FILTER, pheromone.py:8-10
    cfg = cm.make_cfg(E, W_["N"], W_["W"], W_["H"], **extra)
    policy_kind = args.policy or W_.get("policy", "random")
    obs_dtype = args.obs_dtype or ("bf16" if policy_kind == "mlp" else "f32")
    env = BatchedAntsEnv(cfg, dev, obs_dtype=torch.bfloat16 if obs_dtype == "bf16" else torch.float32)
    env.reset(synth_init(cfg, seed=1234, env_offset=rank * E))
    RING = 8
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    rot = torch.randint(-1, 2, (RING, E, cfg.n_ants), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (RING, E, cfg.n_ants), generator=g, device=dev, dtype=torch.int8)
    gather = None
    if dist is not None:
        from antsrl_amd.dist import RewardGather
        gather = RewardGather(world * E, cfg.n_ants, dev)

    policy = None
    if policy_kind == "mlp":
        from antsrl_amd.policy import LinearPolicy
        policy = LinearPolicy(cfg.pside * cfg.pside * cfg.n_channels, dev, seed=5 + rank)
        env.observe()  # main.py:88: first observation feeds the first action

    def one_step(t):
        if gather is not None:
            env.reward, env.done = gather.outputs(t % 2)
        if policy is not None:  # agent.get_action on the device, then api.step + env.update
            a_rot, a_ph = policy.act(env.obs, env.agent_state, env=env)
            env.step_update(a_rot, a_ph, None)
        else:
            env.step_update(rot[t % RING], ph[t % RING], None)
        if gather is not None:
            # the path's only exchange: the reward/done all-gather, one fused
            # collective per step, left running under the next step's kernels.  The step wrote
            # reward/done straight into the gather's send slot (no staging copy).
            gather.start_slot(t % 2)

    for t in range(args.warmup):
        one_step(t)

    K = args.steps
    timing = not args.no_kernel_timing
    # per-kernel HIP events on every 4th step only: three event records between two kernels cost
    # ~15 us of stream idle time (rocprof trace), which would otherwise tax `value` by ~4 %
    EV_EVERY = 4
    timed_steps = list(range(0, K, EV_EVERY))
    evs = HipEvents(4 * len(timed_steps)) if timing else None

    def barrier():
        if gather is not None:
            gather.finish_slot(0)  # the last steps' gathers belong to the timed region
            gather.finish_slot(1)
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    for t in range(K):
        if timing and t % EV_EVERY == 0:
            env.set_timing_events([evs.ev[4 * (t // EV_EVERY) + i].value for i in range(4)])
        one_step(args.warmup + t)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    out = None
    if rank == 0:
        ab = algorithmic_bytes(cfg.n_ants, cfg.w, cfg.h, cfg.n_phero, cfg.n_channels, obs_bytes=2 if obs_dtype == "bf16" else 4)
        kern = {}
        if timing:
            ms = np.array([[evs.elapsed_ms(4 * j + i, 4 * j + i + 1) for i in range(3)] for j in range(len(timed_steps))])
            kern = dict(sweep=float(ms[:, 0].mean()), act=float(ms[:, 1].mean()), update=float(ms[:, 2].mean()))
            if cm.uses_scaled_units(cfg):
                kern.pop("sweep")  # scaled pheromone units: no sweep kernel is launched at all
            dom = max(kern, key=kern.get)
            names = dict(sweep="k_sweep0" if cfg.filter_radius == 0 else "k_sweep_march", act="k_act",
                         update="k_update_one" if cfg.n_ants <= 1024 else "k_update")
            achieved = ab[dom] * E / (kern[dom] * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
            if os.path.exists(tpath):  # PMC-derived HBM bytes per launch, from a separate rocprofv3 --pmc run
                traffic = json.load(open(tpath)).get(names[dom])
            roofline = dict(bound="hbm", kernel=names[dom], achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                            unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                            algorithmic_bytes_per_launch=ab[dom] * E,
                            kernel_ms={names[k]: round(v, 4) for k, v in kern.items()},
                            step_algorithmic_gbs=round(ab["total"] * E / (elapsed / K) / 1e9, 1))
        else:
            roofline = dict(bound="hbm", achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None)
        if timing and roofline.get("achieved"):
            # the spec peak next to what a plain device copy reaches on THIS box
            # (read + write bytes of a 1 GiB float32 copy, outside the timed region)
            copy_gbs = measured_copy_gbs(dev)
            roofline["measured_copy_gbs"] = round(copy_gbs, 1)
            roofline["frac_of_measured_copy"] = round(roofline["achieved"] / copy_gbs, 4)
        value = world * E * cfg.n_ants * K / elapsed
        out = {
            "metric": "ant-steps/sec (ants x envs x steps/s), 256^2 grid" if cfg.w == 256 else "ant-steps/sec (ants x envs x steps/s)",
            "value": value, "unit": "ant-steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 ant kinematics / f32 grids", "data": "synthetic",
            "config": {"workload": W_["desc"], "envs_per_gpu": E, "ants": cfg.n_ants, "grid": [cfg.w, cfg.h],
                       "pheromone_channels": cfg.n_phero, "rocks": cfg.n_rocks, "obs_channels": cfg.n_channels,
                       "filter_radius": cfg.filter_radius, "reward": "ExplorationReward", "obs_dtype": obs_dtype,
                       "pheromone_update": "scaled units (no per-step sweep)" if cm.uses_scaled_units(cfg)
                       else "explicit sweep kernel",
                       "policy": ("linear DQN net (F+2 -> 32 -> 3+3) in-loop, bf16 MFMA" if policy is not None
                                  else "uniform random, pre-generated on device"),
                       "parallelism": "env-sharded x%d, reward/done all-gather" % world},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dict(N=cfg.n_ants, W=cfg.w, H=cfg.h, extra=extra),
                                               cm.make_cfg, synth_init, random_actions)
    if rank == 0 and out is not None:
        # device-side episode reset (antsrl_generate), outside the timed region: how long a whole-batch
        # "EnvironmentGenerator.generate" takes on the GPU
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        env.generate(cm.make_gen(), episode_seed=1)
        torch.cuda.synchronize(dev)
        out["config"]["device_reset_ms"] = round((time.perf_counter() - t1) * 1e3, 3)
    if evs:
        evs.destroy()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
