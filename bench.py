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


GUIDE_COPY_GBS = 6290.0  # float4 copy on MI355X, /opt/skills/guides/MI355X_MICROARCH.md (79 % of the spec peak)


def measured_copy_gbs(dev, nbytes=1 << 30, reps=10):
    """Device-to-device copy bandwidth on this box, GB/s of read + written bytes: the library's own
    16-bytes-per-lane copy kernel (antsrl_bench_copy), not torch's."""
    import torch
    from antsrl_amd import _lib
    lib = _lib.load()
    src = torch.empty((nbytes // 4,), dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def copy():
        _lib.check(lib.antsrl_bench_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), nbytes, st), "bench_copy")
    for _ in range(2):
        copy()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        copy()
    e1.record()
    e1.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def algorithmic_bytes(N, W, H, Cn, K, P=49, obs_bytes=4):
    """SURVEY.md §8(d) per env-step figure, split by the kernel that moves each term.  `move` and `perceive`
    are the two halves of `act` on the cell-meta path (k_move: ant state r+w, actions, food RMW; k_perceive:
    observation, agent_state and reward writes)."""
    sweep = 2 * Cn * W * H * 4 + W * H                 # pheromone read+write sweep, wall mask
    move = N * 60 + N * 8                              # ant x,y,theta f64 r+w, holding, mandibles, actions; food RMW
    perceive = N * P * K * obs_bytes + N * 12          # obs write; agent_state + reward write
    update = N * Cn * 8                                # deposit RMW
    return dict(sweep=sweep, move=move, perceive=perceive, act=move + perceive, update=update,
                total=sweep + move + perceive + update)


from antsrl_amd._lib import HipEvents  # noqa: E402  (raw hipEvent_t handles: the ABI's timing hook records them on the launch stream)


def device_identity(torch, index):
    """What tells one MI355X from another across runs: the device's unique id (the figure `rocm-smi --showuniqueid` prints).
    torch reports it as the device uuid — the id's sixteen hex digits as ASCII bytes; the KFD topology is the fallback."""
    props = None
    try:
        props = torch.cuda.get_device_properties(index)
        raw = bytes.fromhex(str(props.uuid).replace("-", ""))
        txt = raw.decode("ascii")
        int(txt, 16)
        return "0x" + txt.lower()
    except Exception:
        pass
    pci = None
    try:
        pci = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, props.pci_device_id)
    except Exception:
        pass
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        for node in sorted(os.listdir(base)):
            kv = dict(line.split(None, 1) for line in open(os.path.join(base, node, "properties")) if " " in line)
            if int(kv.get("simd_count", "0")) == 0:
                continue  # (a CPU node)
            loc, dom = int(kv.get("location_id", "0")), int(kv.get("domain", "0"))
            if pci is not None and "%04x:%02x:%02x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F) != pci:
                continue
            uid = int(kv.get("unique_id", "0"))
            if uid:
                return "0x%016x" % uid
    except Exception:
        pass
    u = getattr(props, "uuid", None) if props is not None else None
    return str(u) if u is not None else ("pci " + pci if pci else None)


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
    out = dict(value=E * cfg.n_ants * steps / dt, unit="ant-steps/s", cores=cores, kind="port",
               sample="%d envs x %d ants, %dx%d grid, %d full steps (RLApi.step + Environment.update) "
                      "of the same workload; oracle/antsrl_oracle.c, OpenMP over envs, %.1f s"
                      % (E, cfg.n_ants, cfg.w, cfg.h, steps, dt))
    ref = reference_cpu_timing(cfg_kw.get("name"))
    if ref:
        out["reference"] = ref
    return out


def reference_cpu_timing(config_name):
    """The unmodified reference (pure Python + numpy + scipy) cannot travel to the GPU box: it is timed in the
    build container by tests/golden/time_reference.py; the committed record is reported beside the port."""
    path = os.path.join(ROOT, "profiles", "r01", "reference_cpu_timing.json")
    try:
        rec = json.load(open(path))
        key = [k for k in rec["cases"] if k.startswith(config_name + " ")]
        if not key:  # c5 steps c3-shaped environments (no rocks): the c2/c3 shapes bracket it
            return None
        row = rec["cases"][key[0]]
        return dict(value=row["all_core_ant_steps_per_s"], unit="ant-steps/s", cores=row["all_core_processes"],
                    one_process=row["one_process_ant_steps_per_s"], case=key[0],
                    where="build container, %d processes (the reference cannot travel to the GPU box)" % rec["nproc"],
                    source="profiles/r01/reference_cpu_timing.json (tests/golden/time_reference.py)")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--age", type=int, default=400,
                    help="untimed steps BEFORE the warm-up: the reference runs 2000-step episodes (main.py:31) and a step "
                         "costs 10-20 %% more once the ants have spread over the grid (about 150 steps); every timed region, "
                         "kernel average and profile of this command then describes that steady regime, whatever --steps / "
                         "--warmup are (0: time the start-of-episode transient)")
    ap.add_argument("--envs", type=int, default=0, help="override envs per GPU")
    ap.add_argument("--rocks", type=int, default=-1, help="override the number of circle obstacles (profiling)")
    ap.add_argument("--diffuse", type=float, default=0.0,
                    help="the reference's DIFFUSE_FACTOR (pheromone.py:5-10): 3x3 filter with this weight on the 8 neighbours")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--act-path", default="auto", choices=["auto", "meta", "kact"],
                    help="AntsCfg.act_path: auto (the library's measured choice), meta (k_move + k_perceive), kact (round 1's "
                         "single kernel) — for A/B runs on one box")
    ap.add_argument("--gather", default="staged", choices=["staged", "zero_copy", "inline"],
                    help="N > 1: how reward/done reach the all-gather (zero_copy: the kernels write the send slots in place; "
                         "experimental until it has run on RCCL with more than one rank)")
    ap.add_argument("--gather-algo", default="collective", choices=["collective", "direct"],
                    help="N > 1: one all_gather_into_tensor (RCCL picks the algorithm) or SURVEY 8(e)'s one-hop form as a grouped "
                         "batch of sends / receives, every block on the link between its two ranks (RCCL only)")
    ap.add_argument("--no-explicit-sweep", action="store_true",
                    help="skip the short extra run with the per-step sweep kernel forced (explicit_sweep record)")
    ap.add_argument("--policy", default=None, choices=["random", "mlp"],
                    help="mlp: the reference's linear DQN net evaluated in-loop on the GPU (bf16 MFMA)")
    ap.add_argument("--policy-kernel", default="inloop", choices=["inloop", "separate"],
                    help="--policy mlp: the net inside the observation kernel (antsrl_set_inloop_policy; needs bf16 observations) "
                         "or as its own kernel over the observation tensor (antsrl_policy_mlp) — for A/B runs on one box")
    ap.add_argument("--obs-dtype", default=None, choices=["f32", "bf16"],
                    help="observation tensor format (default: f32; c5, whose bf16 policy rounds its input anyway: bf16)")
    ap.add_argument("--obs-row-stride", default=None, choices=["line"],
                    help="opt-in: observation rows a whole number of 128-byte lines apart (antsrl_set_obs_row_stride; measured 8 %% "
                         "SLOWER on c3: profiles/r04/obs_stride_ab_c3.txt) — the default and the headline are the dense tensor")
    ap.add_argument("--no-tune-placement", action="store_true",
                    help="skip BatchedAntsEnv.tune_placement(): by default, before the episode is loaded, up to 8 candidate (workspace, output) "
                         "buffers are stepped on a scratch episode and the fastest physical placement is kept (~0.1 s, outside "
                         "every timed region; DESIGN.md section 2)")
    ap.add_argument("--no-obs", action="store_true",
                    help="--policy mlp, in-loop: act-only rollout (collect_agent_memory.py:189-199 with training=False) — no "
                         "observation tensor is written, the rows feed the net from LDS; rewards / agent_state / done are")
    ap.add_argument("--explicit-sweep", action="store_true",
                    help="force the per-step pheromone sweep kernel (default: scaled units, no sweep)")
    args = ap.parse_args()

    import numpy as np
    import torch

    # ONE JSON line on stdout, whatever the libraries underneath print there (RCCL writes its version banner — five lines — to
    # STDOUT when the process group comes up): file descriptor 1 goes to stderr for the run, the line to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.no_obs and (args.policy or CONFIGS[args.config].get("policy")) != "mlp":
        sys.exit("--no-obs is the act-only rollout of the in-loop policy: use it with --config c5 / --policy mlp")
    if args.gpus > 1 and world == 1:
        sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # ANTSRL_BENCH_BACKEND=gloo + ANTSRL_BENCH_ONE_GPU=1: a one-GPU box's rehearsal of `torch.distributed.run --nproc-per-node N
    # bench.py --gpus N` — every rank on device 0 (RCCL refuses two ranks on one device, so the transport is gloo); everything
    # else is the code the driver's 8-GPU run executes: per-rank shards with their global env ids, tune_placement per rank, the
    # per-step gather, the MAX over ranks, the per-rank arrays of the JSON line.  Not a scaling measurement.
    backend = os.environ.get("ANTSRL_BENCH_BACKEND", "nccl")
    one_gpu = os.environ.get("ANTSRL_BENCH_ONE_GPU") == "1"
    if backend not in ("nccl", "gloo"):
        sys.exit("ANTSRL_BENCH_BACKEND must be nccl (RCCL) or gloo")
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # ANTSRL_BENCH_FORCE_DIST=1: run the N > 1 code path (RCCL process group, reward/done all-gather,
    # max-over-ranks timing) with a single rank — a rehearsal of the multi-GPU launch on a one-GPU box
    force_dist = world == 1 and os.environ.get("ANTSRL_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The per-step all-gather is a small latency-bound collective: on RCCL's default-priority stream it shares a hardware
        # queue with the step's kernels and every step pays two cross-stream hand-overs in series (k_perceive -> gather ->
        # next k_update_move: +41 us per 0.24 ms step, measured with one rank); on a high-priority stream it gets a queue of
        # its own and runs beside the next step (+4 us): profiles/r04/dist_overhead_ab.txt
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # backend "nccl" is RCCL on ROCm
        if not force_dist:  # one rank per GPU, all of them in the process group
            assert dist.get_world_size() == args.gpus == world, \
                "bench.py --gpus %d: the RCCL process group has %d ranks (WORLD_SIZE %d)" % (args.gpus, dist.get_world_size(), world)

    from antsrl_amd import config as cm
    from antsrl_amd.batched import BatchedAntsEnv
    from antsrl_amd.synth import random_actions, synth_init

    W_ = dict(CONFIGS[args.config])
    if args.rocks >= 0:
        W_["R"] = args.rocks
    E = args.envs or W_["E"]
    extra = dict(n_rocks=W_["R"], deposit_strength=256.0, max_time=1 << 30,
                 act_path={"auto": cm.ACT_AUTO, "meta": cm.ACT_CELL_META, "kact": cm.ACT_SINGLE_KERNEL}[args.act_path],
                 phero_mode=cm.PHERO_EXPLICIT_SWEEP if args.explicit_sweep else cm.PHERO_AUTO)
    if W_["radius3"]:
        ax = np.arange(-3, 4)
        g = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 4.5)
        extra["filt"] = g / g.sum() * (1 - 0.001)
    if args.diffuse > 0:
        f3 = np.ones((3, 3)) * args.diffuse
        f3[1, 1] = 1 - 8 * args.diffuse
        extra["filt"] = f3 * (1 - 0.001)  # DIFFUSE_FILTER, pheromone.py:8-10
    # weak scaling: every rank steps E environments of ONE batch of world * E; the rank's block carries its global env ids
    # (AntsCfg.env_id_base), so every environment-keyed random stream — the wall jitter of this loop — is the one the
    # unsharded batch would draw: ranks do not repeat each other's streams and sharding cannot change a result
    env_id_base = rank * E
    extra_id = dict(extra, env_id_base=env_id_base, n_envs_total=world * E)
    cfg = cm.make_cfg(E, W_["N"], W_["W"], W_["H"], **extra_id)
    policy_kind = args.policy or W_.get("policy", "random")
    obs_dtype = args.obs_dtype or ("bf16" if policy_kind == "mlp" else "f32")
    env = BatchedAntsEnv(cfg, dev, obs_dtype=torch.bfloat16 if obs_dtype == "bf16" else torch.float32, obs_row_stride=args.obs_row_stride)
    placement = None if args.no_tune_placement else env.tune_placement()  # (a scratch episode; the real one is loaded next)
    env.reset(synth_init(cfg, seed=1234, env_offset=rank * E))
    RING = 8
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    rot = torch.randint(-1, 2, (RING, E, cfg.n_ants), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (RING, E, cfg.n_ants), generator=g, device=dev, dtype=torch.int8)
    gather = None
    if dist is not None:
        from antsrl_amd.dist import RewardGather
        gather = RewardGather(world * E, cfg.n_ants, dev, algo=args.gather_algo)

    policy, inloop, want_obs = None, False, True
    if policy_kind == "mlp":
        from antsrl_amd.policy import LinearPolicy
        policy = LinearPolicy(cfg.pside * cfg.pside * cfg.n_channels, dev, seed=5 + rank)
        inloop = args.policy_kernel == "inloop" and obs_dtype == "bf16" and bool(env.query(cm.Q_CELL_META))
        if inloop:
            policy.attach(env)  # every observation now also leaves the next actions in env.next_rotation / next_pheromone
        if args.no_obs and not inloop:
            sys.exit("--no-obs needs the in-loop policy (--policy mlp --policy-kernel inloop, bf16 observations)")
        want_obs = not args.no_obs
        env.observe(want_obs=want_obs)  # main.py:88: first observation feeds the first action

    from antsrl_amd.dist import ShardedStepper
    stepper = ShardedStepper(env, gather, args.gather)  # the N > 1 sequence: tests/test_dist_cpu.py runs the same code

    def device_step(t):
        if policy is not None and inloop:  # the actions were computed by the previous observation kernel
            env.step_update(env.next_rotation, env.next_pheromone, None, want_obs=want_obs)
        elif policy is not None:  # agent.get_action on the device, then api.step + env.update
            a_rot, a_ph = policy.act(env.obs, env.agent_state, env=env)
            env.step_update(a_rot, a_ph, None)
        else:
            env.step_update(rot[t % RING], ph[t % RING], None)

    def one_step(t):
        stepper.step(t, lambda: device_step(t))

    # age the episode (not a warm-up: these steps belong to neither W nor K), then W untimed warm-up steps
    for t in range(args.age):
        one_step(t)
    for t in range(args.age, args.age + args.warmup):
        one_step(t)

    K = args.steps
    timing = not args.no_kernel_timing
    REPEATS = max(1, args.repeats)
    NEV = cm.TIMING_EVENTS
    # per-kernel HIP events on every 10th step only: the event records between two kernels cost
    # ~15 us of stream idle time (rocprof trace) — on every step that would tax `value` by ~5 %, on every 10th by 0.5 %
    # (100 sampled steps over the default 5 x 200)
    EV_EVERY = 10
    timed_steps = [(rep, t) for rep in range(REPEATS) for t in range(0, K, EV_EVERY)]  # in EVERY region: the kernel
    # averages then cover the same launches as the step time (and as a rocprofv3 --stats of the same command)
    evs = HipEvents(NEV * len(timed_steps)) if timing else None
    ev_slot = {rt: i for i, rt in enumerate(timed_steps)}

    def barrier():
        """-> (the last step's gathered batch, this rank's clock when ITS work was done).  The last steps' gathers belong to
        the timed region: a rank's clock stops once its kernels AND the collectives it takes part in have completed
        (drain + synchronize).  The closing dist.barrier() then lines the ranks up for the next region; its own latency
        (a 1-element all-reduce plus two host synchronisations, ~0.5 ms = 10 % of a 20-step region) is not part of the K
        steps, and the job's time is the MAX over the ranks of their completion times, all measured from the common start
        behind the opening barrier."""
        gathered = stepper.drain()
        torch.cuda.synchronize(dev)
        t_done = time.perf_counter()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        return gathered, t_done

    gather_checks = 0

    def check_gather(gathered):
        """Once per timed region, outside it: the all-gathered batch holds THIS rank's reward / done of the last step in
        the rows of its global env ids, bit for bit — the first real multi-GPU run checks itself."""
        nonlocal gather_checks
        if gathered is None:
            return
        rew_all, done_all = gathered
        assert tuple(rew_all.shape) == (world * E, cfg.n_ants) and tuple(done_all.shape) == (world * E,), \
            "gathered shapes %s / %s" % (tuple(rew_all.shape), tuple(done_all.shape))
        lo = env_id_base
        assert torch.equal(rew_all[lo:lo + E], env.reward), "rank %d: gathered reward rows [%d, %d) differ from the local tensor" % (rank, lo, lo + E)
        assert torch.equal(done_all[lo:lo + E].to(torch.uint8), env.done), "rank %d: gathered done rows differ from the local tensor" % rank
        gather_checks += 1

    # REPEATS timed regions of exactly K steps each, every one bracketed by barrier + synchronize on both
    # sides and reduced with MAX over the ranks; `value` is the MEDIAN region (median of 5),
    # the spread is reported beside it.
    region_s, region_local_s = [], []
    step_no = args.age + args.warmup
    for rep in range(REPEATS):
        barrier()
        t0 = time.perf_counter()
        for t in range(K):
            if timing and t % EV_EVERY == 0:
                env.set_timing_events([evs.ev[NEV * ev_slot[(rep, t)] + i].value for i in range(NEV)])
            one_step(step_no)
            step_no += 1
        gathered, t_done = barrier()
        el = t_done - t0
        region_local_s.append(el)  # this rank's own clock, before the MAX over ranks
        if dist is not None:
            tmax = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        region_s.append(el)
        check_gather(gathered)
    elapsed = float(np.median(region_s))
    # What the gather costs THIS rank per step: a region with it minus a gather-free region of the same rank, once, outside
    # the timed regions (on an 8-GPU node a slow device and a slow collective must be told apart from the line alone).
    gather_overhead_us = None
    if gather is not None:
        KG = max(10, min(K, 50))
        plain = ShardedStepper(env, None, args.gather)

        def local_region(stp):
            stepper.drain()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for t in range(KG):
                stp.step(step_no + t, lambda: device_step(step_no + t))
            stp.drain()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t1) / KG
        local_region(stepper)  # (warm)
        with_g = min(local_region(stepper) for _ in range(2))
        without_g = min(local_region(plain) for _ in range(2))
        gather_overhead_us = round((with_g - without_g) * 1e6, 2)
        if args.gather in ("zero_copy", "inline"):  # (the plain stepper left reward / done on a send slot: harmless, the run is over)
            pass
    pt = getattr(env, "placement_trials", None) if placement else None
    mine = dict(rank=rank, env_id_base=env_id_base, ms_per_step=float(np.median(region_local_s)) / K * 1e3,
                device=device_identity(torch, local_rank), gather_overhead_us=gather_overhead_us,
                placement_chosen_ms=(pt["ms_per_step"][pt["chosen"]] if pt else None),
                placement_default_ms=(pt["ms_per_step"][0] if pt else None))
    per_rank = [mine]
    if dist is not None:  # every rank's own figures, for the line: the run documents its own sharding and its own devices
        per_rank = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank, mine)
        per_rank.sort(key=lambda r: r["rank"])
    bases = [r["env_id_base"] for r in per_rank]

    out = None
    if rank == 0:
        obs_bytes = (2 if obs_dtype == "bf16" else 4) if want_obs else 0  # act-only: the rows never reach HBM
        ab = algorithmic_bytes(cfg.n_ants, cfg.w, cfg.h, cfg.n_phero, cfg.n_channels, obs_bytes=obs_bytes)
        meta_path = bool(env.query(cm.Q_CELL_META))
        scaled = bool(env.query(cm.Q_SCALED_UNITS))
        deferred = bool(env.query(cm.Q_DEFERRED_UPDATE)) and cfg.n_ants <= 1024
        kern = {}
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        traffic_rec = json.load(open(tpath)) if os.path.exists(tpath) else {}
        if timing:
            ms = np.array([[evs.elapsed_ms(NEV * j + i, NEV * j + i + 1) for i in range(NEV - 1)]
                           for j in range(len(timed_steps))])
            if meta_path and deferred:
                # antsrl_update is deferred into the next step: [1]..[2] brackets k_update_move (the previous step's
                # update + this step's move), [3]..[4] is empty (include/antsrl.h)
                # (under an explicit sweep the step's sweep follows its kernels: [3]..[4], include/antsrl.h)
                kern = dict(sweep=float(ms[:, 0].mean() if scaled else ms[:, 3].mean()), update_move=float(ms[:, 1].mean()),
                            perceive=float(ms[:, 2].mean()))
                ab["update_move"] = ab["move"] + ab["update"]
            elif meta_path:
                kern = dict(sweep=float(ms[:, 0].mean()), move=float(ms[:, 1].mean()), perceive=float(ms[:, 2].mean()),
                            update=float(ms[:, 3].mean()))
            else:
                kern = dict(sweep=float(ms[:, 0].mean()), act=float((ms[:, 1] + ms[:, 2]).mean()), update=float(ms[:, 3].mean()))
            if scaled:
                kern.pop("sweep")  # scaled pheromone units: no sweep kernel is launched at all
            dom = max(kern, key=kern.get)
            sweep_name = ("k_sweep0" if cfg.filter_radius == 0 else
                          "k_sweep_r1x2" if cfg.filter_radius == 1 and cfg.n_phero == 2 and cfg.h % 2 == 0 else
                          "k_sweep_sep2" if cfg.n_phero == 2 and cfg.h % 2 == 0 and env.query(cm.Q_FILTER_SEPARABLE) else
                          "k_sweep_march")
            names = dict(sweep=sweep_name, act="k_act", move="k_move", update_move="k_update_move",
                         perceive="k_perceive", update="k_update_one" if cfg.n_ants <= 1024 else "k_update")
            achieved = ab[dom] * E / (kern[dom] * 1e-3) / 1e9
            # PMC-derived HBM bytes per launch: NOT measured in this run (rocprofv3 --pmc needs its own passes,
            # profiles/history/pmc_traffic.sh); the record names the profile it came from
            traffic = traffic_rec.get(names[dom])
            # the counters are only as current as the tree they were taken on: flagged when the kernel sources have changed since
            from antsrl_amd.build import source_hash
            traffic_stale = traffic_rec.get("_source_sha16") != source_hash() if traffic_rec else None
            roofline = dict(bound="hbm", kernel=names[dom], achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                            unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                            traffic_source=(traffic_rec.get("_source") if traffic is not None else None),
                            traffic_stale=(traffic_stale if traffic is not None else None),
                            traffic_device=(traffic_rec.get("_device") if traffic is not None else None),
                            algorithmic_bytes_per_launch=ab[dom] * E,
                            kernel_ms={names[k]: round(v, 4) for k, v in kern.items()})
            # the WHOLE step against the roofline: the algorithmic bytes of every kernel the step actually launches
            # (no sweep term with scaled units: that pass does not exist) over the measured step time
            step_bytes = sum(ab[k] for k in kern) * E
            roofline["step_algorithmic_bytes"] = step_bytes
            roofline["step_achieved"] = round(step_bytes / (elapsed / K) / 1e9, 1)
            roofline["step_frac"] = round(step_bytes / (elapsed / K) / 1e9 / HBM_PEAK_GBS, 4)
            step_traffic = [traffic_rec.get(names[k]) for k in kern]
            if traffic_rec and all(v is not None for v in step_traffic) and not traffic_stale:
                # (only from counters taken on THIS tree: a stale figure beside a live time would mix two kernels)
                # real HBM bytes of one whole step (PMC, every kernel of the step) over the measured step time
                roofline["step_real_traffic_gbs"] = round(sum(step_traffic) / (elapsed / K) / 1e9, 1)
        else:
            roofline = dict(bound="hbm", achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None)
        if timing and roofline.get("achieved"):
            # the spec peak next to what a plain device copy reaches on THIS box
            # (read + write bytes of a 1 GiB copy, 16 bytes per lane, outside the timed region) and the
            # guide's figure for the same kind of kernel
            copy_gbs = measured_copy_gbs(dev)
            roofline["measured_copy_gbs"] = round(copy_gbs, 1)
            roofline["frac_of_measured_copy"] = round(roofline["achieved"] / copy_gbs, 4)
            roofline["guide_copy_gbs"] = GUIDE_COPY_GBS
            roofline["frac_of_guide_copy"] = round(roofline["achieved"] / GUIDE_COPY_GBS, 4)
        value = world * E * cfg.n_ants * K / elapsed
        out = {
            "metric": "ant-steps/sec (ants x envs x steps/s), 256^2 grid" if cfg.w == 256 else "ant-steps/sec (ants x envs x steps/s)",
            "value": value, "unit": "ant-steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 ant kinematics / f32 grids", "data": "synthetic",
            "repeats": REPEATS, "ms_per_step_regions": [round(r / K * 1e3, 5) for r in region_s],
            "ms_per_step_spread": round((max(region_s) - min(region_s)) / K * 1e3, 5),
            "rccl_ranks": (dist.get_world_size() if dist is not None else 1),  # ranks of the process group (one per GPU)
            "collective_backend": (backend if dist is not None else None),   # "nccl" = RCCL; "gloo" = the one-GPU rehearsal
            "gather": ({"mode": args.gather, "algo": args.gather_algo} if dist is not None else None),
            "env_id_base_per_rank": bases, "n_envs_total": world * E,
            # per-rank figures, rank order: each rank's OWN median region (before the MAX over ranks), its device, what its
            # placement tuner chose (and the default pair's time beside it: the untuned figure), what the gather costs it
            "ms_per_step_per_rank": [round(r["ms_per_step"], 5) for r in per_rank],
            "device_unique_id_per_rank": [r["device"] for r in per_rank],
            "placement_chosen_ms_per_rank": [r["placement_chosen_ms"] for r in per_rank],
            "placement_default_ms_per_rank": [r["placement_default_ms"] for r in per_rank],
            "gather_overhead_us": [r["gather_overhead_us"] for r in per_rank] if dist is not None else None,
            "gather_self_checks": gather_checks,  # regions whose all-gathered own-shard rows were compared with the local tensors
            "config": {"workload": W_["desc"], "episode_age_steps": args.age, "envs_per_gpu": E, "ants": cfg.n_ants, "grid": [cfg.w, cfg.h],
                       "pheromone_channels": cfg.n_phero, "rocks": cfg.n_rocks, "obs_channels": cfg.n_channels,
                       "filter_radius": cfg.filter_radius,
                       "filter_separable": bool(env.query(cm.Q_FILTER_SEPARABLE)) if cfg.filter_radius else None,
                       "obs_row_pitch_elems": env.obs_row_pitch,
                       "placement_trials_ms_per_step": (getattr(env, "placement_trials", None) if placement else None), "reward": "ExplorationReward", "obs_dtype": obs_dtype if want_obs else "none (act-only: rows stay in LDS)",
                       "pheromone_update": "scaled units (no per-step sweep)" if scaled else "explicit sweep kernel",
                       "kernels": ("k_update_move (the previous step's update + this step's move, one launch) + k_perceive "
                                   "(cell-meta layout, %d ants per wave)" % env.query(cm.Q_PERCEIVE_RUN)) if meta_path and deferred
                       else (("k_move + k_perceive (cell-meta layout, %d ants per wave) + " % env.query(cm.Q_PERCEIVE_RUN)
                              if meta_path else "k_act + ") + ("k_update_one" if cfg.n_ants <= 1024 else "k_update")),
                       "policy": (("linear DQN net (F+2 -> 32 -> 3+3) in-loop, bf16 MFMA, " +
                                   ("inside k_perceive (antsrl_set_inloop_policy)" if inloop else "own kernel (antsrl_policy_mlp)"))
                                  if policy is not None
                                  else "uniform random, pre-generated on device"),
                       "parallelism": "env-sharded x%d, reward/done all-gather (%s)" % (world, args.gather)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dict(N=cfg.n_ants, W=cfg.w, H=cfg.h, extra=extra, name=args.config),
                                               cm.make_cfg, synth_init, random_actions)
    if rank == 0 and out is not None and world == 1 and scaled and not args.no_explicit_sweep and policy is None:
        # The same workload with the per-step sweep kernel forced (SURVEY.md §8(d)'s byte model counts the
        # 2*C*W*H*4 + W*H sweep term, which the scaled units never move): a short run of its own so that the
        # model can be checked like for like.  Outside the timed region.
        del env
        torch.cuda.empty_cache()
        ex = dict(extra_id)
        ex["phero_mode"] = cm.PHERO_EXPLICIT_SWEEP
        cfg_x = cm.make_cfg(E, W_["N"], W_["W"], W_["H"], **ex)
        env_x = BatchedAntsEnv(cfg_x, dev)
        env_x.reset(synth_init(cfg_x, seed=1234, env_offset=rank * E))
        for t in range(args.age + 10):  # (the same age as the main run)
            env_x.step_update(rot[t % RING], ph[t % RING], None)
        KX, EVX = 60, 5
        evx = HipEvents(NEV * (KX // EVX))
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for t in range(KX):
            if t % EVX == 0:
                env_x.set_timing_events([evx.ev[NEV * (t // EVX) + i].value for i in range(NEV)])
            env_x.step_update(rot[t % RING], ph[t % RING], None)
        torch.cuda.synchronize(dev)
        ms_x = (time.perf_counter() - t1) / KX * 1e3
        # ([0]..[1], or — behind a deferred update — [3]..[4]: the other bracket is empty, include/antsrl.h)
        sweep_ms = float(np.mean([evx.elapsed_ms(NEV * j, NEV * j + 1) + evx.elapsed_ms(NEV * j + 3, NEV * j + 4) for j in range(KX // EVX)]))
        evx.destroy()
        out["explicit_sweep"] = dict(ms_per_step=round(ms_x, 4), k_sweep0_ms=round(sweep_ms, 4),
                                     k_sweep0_algorithmic_gbs=round(ab["sweep"] * E / (sweep_ms * 1e-3) / 1e9, 1),
                                     step_algorithmic_gbs=round(ab["total"] * E / (ms_x * 1e-3) / 1e9, 1),
                                     note="same workload with ANTSRL_PHERO_EXPLICIT_SWEEP: the configuration SURVEY.md "
                                          "§8(d)'s 1 865 728 B/env-step model describes")
        env = env_x
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
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
