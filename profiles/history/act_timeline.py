"""Workgroup phase timeline of k_act (profiling; sets ANTSRL_ABLATE=32768 = ACT_ABL_TRACE).

    python3 profiles/act_timeline.py [--envs E] [--out gpurun_out/act_timeline.json]

Runs the c3 workload for a few steps, reads the per-workgroup stamps of the LAST k_act launch
(antsrl_debug_read_act_trace) and prints when workgroups start, how long their per-ant phases
(0-2) and perception (3) take in wall-clock, and how they spread over XCDs / CUs."""
import argparse
import ctypes as C
import json
import os
import sys

os.environ["ANTSRL_ABLATE"] = str(int(os.environ.get("ANTSRL_ABLATE", "0")) | 32768)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from antsrl_amd import _lib
from antsrl_amd import config as cm
from antsrl_amd.batched import BatchedAntsEnv
from antsrl_amd.synth import synth_init


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--ants", type=int, default=512)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    E = a.envs
    cfg = cm.make_cfg(E, a.ants, 256, 256, n_rocks=8, deposit_strength=256.0, max_time=1 << 30)
    env = BatchedAntsEnv(cfg, dev)
    env.reset(synth_init(cfg, seed=1234))
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    rot = torch.randint(-1, 2, (4, E, cfg.n_ants), generator=g, device=dev, dtype=torch.int8)
    ph = torch.randint(0, 3, (4, E, cfg.n_ants), generator=g, device=dev, dtype=torch.int8)
    for t in range(12):
        env.step_update(rot[t % 4], ph[t % 4], None)
    torch.cuda.synchronize()
    lib = _lib.load()
    buf = np.zeros((E, 8), np.uint64)
    rc = lib.antsrl_debug_read_act_trace(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), E)
    assert rc == 0, rc
    t = buf[:, :4].astype(np.int64)
    t0 = t.min()
    us = (t - t0) / 100.0  # 100 MHz
    hw = buf[:, 4]
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    xcc = buf[:, 5] & 0xF
    cu_uid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    res = dict(
        envs=E, span_us=float(us[:, 3].max()),
        start_us=dict(p50=float(np.median(us[:, 0])), p90=float(np.percentile(us[:, 0], 90)), max=float(us[:, 0].max())),
        phases012_us=dict(mean=float((us[:, 1] - us[:, 0]).mean()), p90=float(np.percentile(us[:, 1] - us[:, 0], 90))),
        perception_us=dict(mean=float((us[:, 2] - us[:, 1]).mean()), p90=float(np.percentile(us[:, 2] - us[:, 1], 90))),
        phase0_us=dict(mean=float(((buf[:, 6].astype(np.int64) - t0) / 100.0 - us[:, 0]).mean())),
        phase1_us=dict(mean=float(((buf[:, 7].astype(np.int64) - buf[:, 6].astype(np.int64)) / 100.0).mean())),
        phase2_us=dict(mean=float((us[:, 1] - (buf[:, 7].astype(np.int64) - t0) / 100.0).mean())),
        tail_us=dict(mean=float((us[:, 3] - us[:, 2]).mean())),
        wg_per_xcc=np.bincount(xcc.astype(np.int64), minlength=8).tolist(),
        distinct_cus=int(len(np.unique(cu_uid))),
        wg_per_cu=dict(min=int(np.bincount(np.unique(cu_uid, return_inverse=True)[1]).min()),
                       max=int(np.bincount(np.unique(cu_uid, return_inverse=True)[1]).max())),
    )
    # how many workgroups are inside their perception phase at each 10 us tick
    ticks = np.arange(0, us[:, 3].max() + 10, 10.0)
    res["in_perception_per_10us"] = [int(((us[:, 1] <= x) & (us[:, 2] > x)).sum()) for x in ticks]
    res["in_phases012_per_10us"] = [int(((us[:, 0] <= x) & (us[:, 1] > x)).sum()) for x in ticks]
    late = us[:, 0] > 0.25 * us[:, 3].max()
    res["late_starters"] = dict(count=int(late.sum()),
                                perception_us=float((us[late, 2] - us[late, 1]).mean()) if late.any() else None,
                                early_perception_us=float((us[~late, 2] - us[~late, 1]).mean()))
    print(json.dumps(res))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
