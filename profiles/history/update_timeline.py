"""Phase timeline of k_update_one (instrumented build, not in the tree: see profiles/update_timeline.patch):

    git apply profiles/update_timeline.patch && AB_FLAGS=-DANTSRL_AB_UTRACE profiles/ab.sh build utrace && git checkout antsrl_amd/csrc/antsrl_update.hip
    gpurun -- 'ANTSRL_LIB=$GRAFT_REPO_ROOT/antsrl_amd/lib/variants/utrace.so python3 profiles/update_timeline.py'

Each workgroup's last thread stamps the shader clock at the phase boundaries; thread 0 stamps the 100 MHz
wall clock at entry and exit (dispatch skew between workgroups)."""
import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from antsrl_amd import _lib, config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init
E = 1024
cfg = cm.make_cfg(E, 512, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
dev = torch.device("cuda", 0); env = BatchedAntsEnv(cfg, dev); env.reset(synth_init(cfg, seed=1234))
g = torch.Generator(device=dev); g.manual_seed(99)
rot = torch.randint(-1, 2, (4, E, cfg.n_ants), generator=g, device=dev, dtype=torch.int8)
ph = torch.randint(0, 3, (4, E, cfg.n_ants), generator=g, device=dev, dtype=torch.int8)
for t in range(40): env.step_update(rot[t % 4], ph[t % 4], None)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["ANTSRL_LIB"])
buf = np.zeros((4096, 16), np.uint64)
assert lib.antsrl_debug_read_upd_trace(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 4096 * 16) == 0
b = buf[:E].astype(np.int64)
names = ["entry", "loads issued+LDS init", "walls bit", "jitter/revert", "sx/sy + barrier", "rock pass 1 + barrier", "rock pass 2",
         "barrier (hash ready)", "state stores + hash insert", "barrier", "walldep clear", "deposit + collect", "reduce + barrier", "tail"]
d = np.diff(b[:, :14], axis=1)
print("shader-clock cycles per phase (mean / p90 over %d workgroups):" % E)
for k in range(13):
    print("  %-28s %8.0f %8.0f" % (names[k + 1], d[:, k].mean(), np.percentile(d[:, k], 90)))
tot = b[:, 13] - b[:, 0]
print("  total %.0f cycles mean, %.0f p90" % (tot.mean(), np.percentile(tot, 90)))
w0, w1 = b[:, 14], b[:, 15]
print("wall clock (us): workgroup lifetime mean %.2f p90 %.2f; first entry -> last exit %.2f; entry skew p50 %.2f p99 %.2f" % (
    (w1 - w0).mean() / 100, np.percentile(w1 - w0, 90) / 100, (w1.max() - w0.min()) / 100,
    np.percentile(w0 - w0.min(), 50) / 100, np.percentile(w0 - w0.min(), 99) / 100))
