"""two_stream_probe.py with a per-step fork / join (the second half is released after the first half's whole step call
returns to its stream, or at once) — does the gain of two free-running streams survive the join a library-internal
split needs?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init

dev = torch.device("cuda:0")
N, W, H, R = 512, 256, 256, 8


def make(E, off):
    cfg = cm.make_cfg(E, N, W, H, n_rocks=R, deposit_strength=256.0, max_time=1 << 30)
    env = BatchedAntsEnv(cfg, dev)
    env.reset(synth_init(cfg, seed=1234, env_offset=off))
    g = torch.Generator(device=dev); g.manual_seed(99 + off)
    rot = torch.randint(-1, 2, (8, E, N), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (8, E, N), generator=g, device=dev, dtype=torch.int8)
    return env, rot, ph


two = [make(512, 0), make(512, 512)]
main = torch.cuda.current_stream(dev)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def run(mode, steps=400, warm=100):
    for t in range(warm + steps):
        if t == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        (ea, ra, pa), (eb, rb, pb) = two
        if mode == "free":
            with torch.cuda.stream(sa): ea.step_update(ra[t % 8], pa[t % 8], None)
            with torch.cuda.stream(sb): eb.step_update(rb[t % 8], pb[t % 8], None)
        elif mode == "join":  # fork from main, both halves at once, join on main
            f = torch.cuda.Event(); f.record(main)
            sa.wait_event(f); sb.wait_event(f)
            with torch.cuda.stream(sa): ea.step_update(ra[t % 8], pa[t % 8], None)
            with torch.cuda.stream(sb): eb.step_update(rb[t % 8], pb[t % 8], None)
            ja, jb = torch.cuda.Event(), torch.cuda.Event(); ja.record(sa); jb.record(sb)
            main.wait_event(ja); main.wait_event(jb)
        elif mode == "main+helper":  # half A on main itself, half B on a helper forked at the start, joined at the end
            f = torch.cuda.Event(); f.record(main)
            sb.wait_event(f)
            ea.step_update(ra[t % 8], pa[t % 8], None)
            with torch.cuda.stream(sb): eb.step_update(rb[t % 8], pb[t % 8], None)
            jb = torch.cuda.Event(); jb.record(sb); main.wait_event(jb)
        elif mode == "one":
            ea.step_update(ra[t % 8], pa[t % 8], None); eb.step_update(rb[t % 8], pb[t % 8], None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for mode in ("free", "join", "main+helper", "one", "free"):
    print("%-12s %.4f ms/step" % (mode, run(mode)))
