"""Time line of k_update_move's workgroups (variant build -DUM_TRACE: antsrl_device.h, antsrl_update_one.h, antsrl_perceive.hip).
   python3 profiles/um_trace.py [c3|c2|c4|c5]     (on a GPU box; needs antsrl_amd/lib/variants/umtrace.so:
                                                   python -m antsrl_amd.build --variant umtrace -DUM_TRACE)
Stamps (thread 0 of every workgroup, 10 ns ticks): 0 entry, 1 first loads issued, 2 wall bits back (x / y + the dependent
bit-map word), 3 behind the rock pass's first barrier, 4 pass 1 done (barrier), 5 pass 2 done, 6 record load issued + hash
inserts (barrier), 7 wall-deposit clear (barrier), 8 deposit / collect stores issued, 9 end of the update (2 barriers),
10 move: first barrier, 11 mandibles + food hash (barrier), 12 food exchange, 13 sincos / move / presence stamp issued,
14 every store acknowledged (trace build only), 15 HW_ID."""
import os, sys, ctypes as C
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
os.environ["ANTSRL_LIB"] = os.path.join(R, "antsrl_amd/lib/variants/%s.so" % os.environ.get("TRACE_VARIANT", "umtrace"))
sys.path.insert(0, R)
import numpy as np, torch
from antsrl_amd import _lib, config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(R, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
W_ = bench.CONFIGS[which]
E, N, W = W_["E"], W_["N"], W_["W"]
dev = torch.device("cuda", 0)
kw = dict(n_rocks=W_["R"], deposit_strength=256.0, max_time=1 << 30)
if W_["radius3"]:  # (bench.py's filter for c4)
    ax = np.arange(-3, 4)
    gf = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) / 4.5)
    kw["filt"] = gf / gf.sum() * (1 - 0.001)
cfg = cm.make_cfg(E, N, W, W_["H"], **kw)
mlp = W_.get("policy") == "mlp"
env = BatchedAntsEnv(cfg, dev, obs_dtype=torch.bfloat16 if mlp else torch.float32); env.reset(synth_init(cfg, seed=1234))
g = torch.Generator(device=dev); g.manual_seed(99)
rot = torch.randint(-1, 2, (8, E, N), generator=g, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (8, E, N), generator=g, device=dev, dtype=torch.int8)
if mlp:
    from antsrl_amd.policy import LinearPolicy
    LinearPolicy(cfg.pside * cfg.pside * cfg.n_channels, dev, seed=5).attach(env)
    env.observe()
age = int(os.environ.get("AGE", "400"))
for t in range(age):
    if mlp: env.step_update(env.next_rotation, env.next_pheromone, None)
    else: env.step_update(rot[t % 8], ph[t % 8], None)
torch.cuda.synchronize()
S = 16
buf = np.zeros((E, S), np.uint32)
lib = _lib.load()
lib.antsrl_debug_read_um_trace.argtypes = [C.POINTER(C.c_uint32), C.c_int]
assert lib.antsrl_debug_read_um_trace(buf.ctypes.data_as(C.POINTER(C.c_uint32)), E) == 0
t = buf[:, :15].astype(np.int64)
ok = t[:, 0] > 0
t = t[ok]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
span = us[:, 14].max()
print("%s: %d envs x %d ants, %d rocks: %d workgroups traced, kernel span %.1f us (first entry -> last store acknowledged)" % (which, E, N, W_["R"], len(t), span))
names = ["entry -> first loads issued", "-> wall bits back (2 dependent round trips)", "-> rock positions in LDS (barrier)", "-> rock pass 1 (barrier)",
         "-> rock pass 2", "-> record load issued, hash inserts (barrier)", "-> wall-deposit clear (barrier)", "-> deposit + collect stores issued",
         "-> end of the update (reduction, 2 barriers)", "-> move: table init (barrier)", "-> mandibles, food hash (barrier)", "-> food exchange",
         "-> rotate, sincos, move, stamp issued", "-> stores acknowledged (trace build only)"]
has_rocks = W_["R"] > 0
entry = us[:, 0]
first = entry < np.median(entry) - 1e-9 if (entry.max() - entry.min()) > 5 else np.ones(len(us), bool)
for label, m in (("first round (workgroups that enter before the median entry time)", first), ("later workgroups", ~first)):
    if m.sum() == 0: continue
    print("%s: %d, entry at %.1f .. %.1f us" % (label, m.sum(), entry[m].min(), entry[m].max()))
    prev = 0
    for k in range(1, 15):
        if not has_rocks and k in (3, 4, 5): continue
        if (us[m, k] <= 0).all() and (t[m, k] == 0).all(): continue
        d = us[m, k] - us[m, prev]
        print("  %2d %-52s mean %6.2f us  median %6.2f  p90 %6.2f" % (k, names[k - 1], d.mean(), np.median(d), np.percentile(d, 90)))
        prev = k
    life = us[m, 13] - us[m, 0]
    print("     %-52s mean %6.2f us  median %6.2f  p90 %6.2f" % ("workgroup: entry -> last instruction issued", life.mean(), np.median(life), np.percentile(life, 90)))
ev = np.concatenate([np.stack([us[:, 0], np.ones(len(us))], 1), np.stack([us[:, 14], -np.ones(len(us))], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1])
for frac in (0.1, 0.3, 0.5, 0.7, 0.9):
    i = np.searchsorted(ev[:, 0], frac * span)
    print("  workgroups alive at %2.0f %% of the span: %d (%.2f per CU)" % (100 * frac, alive[min(i, len(alive) - 1)], alive[min(i, len(alive) - 1)] / 256))
