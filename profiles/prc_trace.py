"""Time line of k_perceive's waves (variant build -DPRC_TRACE, antsrl_perceive.hip): where does a workgroup's life go?
   python3 profiles/prc_trace.py [c3|c2|c4|c5|c5a]   (c5a: c5 act-only)      (on a GPU box; builds nothing: needs antsrl_amd/lib/variants/trace.so)
Stamps (10 ns ticks): 0 entry, 1 past the prologue's barrier, 2 first gathers back, 3..6 behind group 1..4 (stores issued),
7 loop done, 8 every store acknowledged (the trace build waits for them; the product does not); in-loop policy (c5 / c5a):
11 behind the barrier in front of the net, 12 (wave 0 only) the net's actions are out."""
import os, sys, ctypes as C
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
os.environ["ANTSRL_LIB"] = os.path.join(R, "antsrl_amd/lib/variants/%s.so" % os.environ.get("TRACE_VARIANT", "trace"))
os.environ["ANTSRL_PRC_LDS_PAD"] = "1"
sys.path.insert(0, R)
import numpy as np, torch
from antsrl_amd import _lib, config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(R, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
W_ = bench.CONFIGS[which.rstrip("a")]
E, N, W = W_["E"], W_["N"], W_["W"]
dev = torch.device("cuda", 0)
cfg = cm.make_cfg(E, N, W, W_["H"], n_rocks=W_["R"], deposit_strength=256.0, max_time=1 << 30)  # (bench.py's configuration)
mlp = W_.get("policy") == "mlp"
env = BatchedAntsEnv(cfg, dev, obs_dtype=torch.bfloat16 if mlp else torch.float32); env.reset(synth_init(cfg, seed=1234))
g = torch.Generator(device=dev); g.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=g, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (8, E, N), generator=g, device=dev, dtype=torch.int8)
want_obs = not which.endswith("a")
if mlp:
    from antsrl_amd.policy import LinearPolicy
    LinearPolicy(cfg.pside * cfg.pside * cfg.n_channels, dev, seed=5).attach(env)
    env.observe(want_obs=want_obs)
age = int(os.environ.get("AGE", "400"))
for t in range(age):
    if mlp: env.step_update(env.next_rotation, env.next_pheromone, None, want_obs=want_obs)
    else: env.step_update(rot[t % 8], ph[t % 8], None)
torch.cuda.synchronize()
nw = E * ((N + 31) // 32) * 4
buf = np.zeros((nw, 16), np.uint32)
lib = _lib.load()
lib.antsrl_debug_read_prc_trace.argtypes = [C.POINTER(C.c_uint32), C.c_int]
assert lib.antsrl_debug_read_prc_trace(buf.ctypes.data_as(C.POINTER(C.c_uint32)), nw) == 0
t = buf[:, :9].astype(np.int64)
ok = t[:, 0] > 0
t = t[ok]; hw = buf[ok, 9]; xcc = buf[ok, 10]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
span = us[:, 8].max()
print("%s: %d envs x %d ants, %d waves traced, kernel span %.1f us" % (which, E, N, len(t), span))
mid = (us[:, 0] > 0.2 * span) & (us[:, 0] < 0.7 * span)
names = ["prologue (entry -> past the barrier)", "first gathers back", "group 1 (count, stage, copy-out issued)", "group 2", "group 3", "group 4",
         "loop exit", "stores acknowledged (trace build only)"]
print("waves that start in the middle of the launch (%d):" % mid.sum())
for k in range(8):
    d = us[mid, k + 1] - us[mid, k]
    print("  %-46s mean %6.2f us  median %6.2f  p90 %6.2f" % (names[k], d.mean(), np.median(d), np.percentile(d, 90)))
life = us[mid, 7] - us[mid, 0]
print("  %-46s mean %6.2f us  median %6.2f  p90 %6.2f" % ("wave: entry -> loop exit", life.mean(), np.median(life), np.percentile(life, 90)))
wg = us.reshape(-1, 4, 9) if len(us) % 4 == 0 and ok.all() else None
if wg is not None:
    wl = wg[:, :, 7].max(1) - wg[:, :, 0].min(1)
    m = (wg[:, :, 0].min(1) > 0.2 * span) & (wg[:, :, 0].min(1) < 0.7 * span)
    print("  %-46s mean %6.2f us  median %6.2f  p90 %6.2f" % ("workgroup: first entry -> last loop exit", wl[m].mean(), np.median(wl[m]), np.percentile(wl[m], 90)))
    print("  workgroups resident per CU (sum of lives / span / 256): %.2f" % (wl.sum() / span / 256))
    skew = wg[:, :, 7].max(1) - wg[:, :, 7].min(1)
    print("  spread of the four waves' loop exits inside a workgroup: mean %.2f us" % skew[m].mean())
if mlp and wg is not None:  # the in-loop net: one wave per workgroup evaluates the tile's 32 rows behind a workgroup barrier
    pt = (buf[ok, 11].astype(np.int64) - t0) / 100.0
    pe = (buf[ok, 12].astype(np.int64) - t0) / 100.0
    ptw, pew = pt.reshape(-1, 4), pe.reshape(-1, 4)
    wait = ptw - wg[:, :, 8]          # every wave: stores acknowledged -> past the barrier (waiting for the slowest wave)
    net = pew[:, 0] - ptw[:, 0]       # wave 0: the net
    full = pew[:, 0] - wg[:, :, 0].min(1)
    print("  %-46s mean %6.2f us  median %6.2f  p90 %6.2f" % ("in-loop net: wait at its barrier (all waves)", wait[m].mean(), np.median(wait[m]), np.percentile(wait[m], 90)))
    print("  %-46s mean %6.2f us  median %6.2f  p90 %6.2f" % ("in-loop net: wave 0 alone (MFMA tile + argmax)", net[m].mean(), np.median(net[m]), np.percentile(net[m], 90)))
    print("  %-46s mean %6.2f us  median %6.2f  p90 %6.2f" % ("workgroup: first entry -> actions out", full[m].mean(), np.median(full[m]), np.percentile(full[m], 90)))
    print("  workgroups resident per CU incl. the net's tail (sum of full lives / span / 256): %.2f" % (full.sum() / max(span, pew[:, 0].max()) / 256))
# how many waves are alive over time
ev = np.concatenate([np.stack([us[:, 0], np.ones(len(us))], 1), np.stack([us[:, 7], -np.ones(len(us))], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1])
for frac in (0.1, 0.3, 0.5, 0.7, 0.9):
    i = np.searchsorted(ev[:, 0], frac * span)
    print("  waves alive at %2.0f %% of the span: %d (%.1f per CU)" % (100 * frac, alive[min(i, len(alive) - 1)], alive[min(i, len(alive) - 1)] / 256))
