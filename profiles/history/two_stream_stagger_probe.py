"""Staggered half-batches with a per-step join (round 3): two handles of 512 envs; stream S steps half A, stream H steps half B
but only starts once A's k_update_move is done (the library's timing hook records an event there), so B's latency-bound
k_update_move runs beside A's streaming k_perceive; S then waits for H (join) — the contract of one step_update call stays.
Compared in ONE process on the SAME allocations with: both halves on one stream; both on two streams with fork + join but no
stagger; free-running two streams (no join); and one handle of 1024 envs (other allocations: placement differs)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from antsrl_amd import _lib, config as cm  # noqa: E402
from antsrl_amd.batched import BatchedAntsEnv  # noqa: E402
from antsrl_amd.synth import synth_init  # noqa: E402

dev = torch.device("cuda:0")
N, W, H_, R = 512, 256, 256, 8
torch.zeros(1, device=dev)
hip = _lib.hip_runtime()
hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]


def make(E, off):
    cfg = cm.make_cfg(E, N, W, H_, n_rocks=R, deposit_strength=256.0, max_time=1 << 30)
    env = BatchedAntsEnv(cfg, dev)
    env.reset(synth_init(cfg, seed=1234, env_offset=off))
    g = torch.Generator(device=dev)
    g.manual_seed(99 + off)
    rot = torch.randint(-1, 2, (8, E, N), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (8, E, N), generator=g, device=dev, dtype=torch.int8)
    return env, rot, ph


def raw_events(n):
    out = []
    for _ in range(n):
        e = C.c_void_p()
        assert hip.hipEventCreateWithFlags(C.byref(e), 0x2) == 0  # hipEventDisableTiming
        out.append(e.value)
    return out


def run(mode, A, B, S, Hs, steps, warm):
    (ea, ra, pa), (eb, rb, pb) = A, B
    evs = raw_events(cm.TIMING_EVENTS)
    join, fork = torch.cuda.Event(), torch.cuda.Event()
    for t in range(warm + steps):
        if t == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if mode == "one stream":
            with torch.cuda.stream(S):
                ea.step_update(ra[t % 8], pa[t % 8], None)
                eb.step_update(rb[t % 8], pb[t % 8], None)
        elif mode == "free-running":
            with torch.cuda.stream(S):
                ea.step_update(ra[t % 8], pa[t % 8], None)
            with torch.cuda.stream(Hs):
                eb.step_update(rb[t % 8], pb[t % 8], None)
        else:
            with torch.cuda.stream(S):
                fork.record(S)
                if mode == "staggered + join":
                    ea.set_timing_events(evs)  # evs[2] is recorded behind A's k_update_move
                ea.step_update(ra[t % 8], pa[t % 8], None)
            if mode == "staggered + join":
                assert hip.hipStreamWaitEvent(C.c_void_p(Hs.cuda_stream), C.c_void_p(evs[2]), 0) == 0
            else:
                Hs.wait_event(fork)
            with torch.cuda.stream(Hs):
                eb.step_update(rb[t % 8], pb[t % 8], None)
                join.record(Hs)
            S.wait_event(join)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


S, Hs = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
A, B = make(512, 0), make(512, 512)
for rep in range(2):
    for mode in ("one stream", "fork + join", "staggered + join", "free-running"):
        print("%-18s %.4f ms/step" % (mode, run(mode, A, B, S, Hs, 400, 420 if rep == 0 and mode == "one stream" else 20)), flush=True)
del A, B
torch.cuda.empty_cache()
one = make(1024, 0)
for rep in range(2):
    t = run("one stream", one, (one[0], one[1], one[2]), S, Hs, 0, 0) if False else None
env, rot, ph = one
for t in range(420):
    env.step_update(rot[t % 8], ph[t % 8], None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(400):
    env.step_update(rot[t % 8], ph[t % 8], None)
torch.cuda.synchronize()
print("one handle x 1024   %.4f ms/step (other allocations)" % ((time.perf_counter() - t0) / 400 * 1e3))
