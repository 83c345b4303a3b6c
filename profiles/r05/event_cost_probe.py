#!/usr/bin/env python3
"""What an event record / a cross-stream hand-over costs the step's stream, by event flags (c3 loop, one device):
    python3 profiles/r05/event_cost_probe.py
A: nothing; B: a torch.cuda.Event recorded on the step's stream every step; C: a raw HIP event (DisableTiming) recorded every
step; D: the same with hipEventDisableSystemFence; E: hand-over to a side stream and back two steps later with torch events
(what a collective on its own stream does); F: the same hand-over with fence-free raw events."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from antsrl_amd import config as cm, _lib
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
hip = _lib.hip_runtime()
hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]


def raw_events(n, flags):
    out = []
    for _ in range(n):
        e = C.c_void_p()
        assert hip.hipEventCreateWithFlags(C.byref(e), flags) == 0
        out.append(e)
    return out


DT, NOFENCE, TODEV = 0x2, 0x20000000, 0x40000000
side = torch.cuda.Stream(device=dev, priority=-1)
small = torch.zeros(1 << 19, device=dev)  # 2 MiB: the size of the gather's payload
small2 = torch.zeros_like(small)
STEPS = 100


def loop(kind):
    main = torch.cuda.current_stream(dev)
    ms, ss = C.c_void_p(main.cuda_stream), C.c_void_p(side.cuda_stream)
    tev = [torch.cuda.Event() for _ in range(4)]
    flags = {"C": DT, "D": DT | NOFENCE, "F": DT | NOFENCE, "G": DT | TODEV}.get(kind, DT)
    rev = raw_events(4, flags)
    back_t = [None, None]
    back_r = [False, False]
    for t in range(20):
        env.step_update(rot[t % 8], ph[t % 8], None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(STEPS):
        k = t & 1
        env.step_update(rot[t % 8], ph[t % 8], None)
        if kind == "B":
            tev[k].record(main)
        elif kind in ("C", "D", "G"):
            hip.hipEventRecord(rev[k], ms)
        elif kind == "E":
            if back_t[k] is not None:
                main.wait_event(back_t[k])
            tev[k].record(main)
            side.wait_event(tev[k])
            with torch.cuda.stream(side):
                small2.copy_(small)
                tev[2 + k].record(side)
            back_t[k] = tev[2 + k]
        elif kind == "F":
            if back_r[k]:
                hip.hipStreamWaitEvent(ms, rev[2 + k], 0)
            hip.hipEventRecord(rev[k], ms)
            hip.hipStreamWaitEvent(ss, rev[k], 0)
            with torch.cuda.stream(side):
                small2.copy_(small)
            hip.hipEventRecord(rev[2 + k], ss)
            back_r[k] = True
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


for t in range(400):
    env.step_update(rot[t % 8], ph[t % 8], None)
names = dict(A="nothing", B="torch event record per step", C="raw event (DisableTiming) record per step",
             D="raw event (DisableTiming | DisableSystemFence) record per step", G="raw event (DisableTiming | ReleaseToDevice) record per step",
             E="hand-over to a side stream and back, torch events", F="hand-over to a side stream and back, fence-free raw events")
for rep in range(2):
    base = None
    for kind in "ABCDGEF":
        ms = loop(kind)
        base = ms if kind == "A" else base
        print("%s  %-70s %.4f ms/step  (%+.1f us)" % (kind, names[kind], ms, (ms - base) * 1e3), flush=True)
