"""The N > 1 path on real GPUs over RCCL: runs only where the box has at least two GPUs (the builder's
boxes have one: skipped there; tests/test_dist_cpu.py covers the same code over gloo).  Two fresh child processes are
started by torch.distributed.run — the parent never shares its GPU context with them — and each checks the sharded,
all-gathered reward / done of every step against a single-rank run of the whole batch, in `staged` and `zero_copy` mode
(tests/dist_gpu_worker.py).

The file name sorts FIRST on purpose: the children are started by fork + exec, and on the GPU pool a process that has
initialised the GPU must not exec — pytest collects files in name order, so these tests run before any other test has
touched the GPU in the pytest process (torch.cuda.device_count() does not initialise it).  Only a box with fewer GPUs than
ranks or a REFUSED spawn (OSError from the pool's exec guard) skips; a worker that runs and fails fails the test."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, env):
    try:
        return subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    except OSError as e:  # the pool's exec guard
        pytest.skip("child processes cannot be started here: %s" % e)


@pytest.mark.parametrize("world", [2])
def test_sharded_stepper_over_rccl(world):
    import torch
    n = torch.cuda.device_count()  # (does not initialise the GPU on this image)
    if n < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, n))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    out = _run(cmd, env)
    ok = out.returncode == 0 and all("DIST_GPU_OK %d of %d" % (r, world) in out.stdout for r in range(world))
    log = os.path.join(ROOT, "gpurun_out", "dist_gpu_%d_ranks.log" % world)
    if os.path.isdir(os.path.dirname(log)):
        open(log, "w").write(out.stdout)
    # A hard assertion: every shard carries its global environment ids (AntsCfg.env_id_base), so the library's own
    # wall jitter is the whole batch's and a difference here is a real N > 1 bug.  The worker's output is in gpurun_out/ too.
    assert ok, "two RCCL ranks: the worker failed (rc %d)\n%s" % (out.returncode, out.stdout[-6000:])


def test_worker_runs_with_one_rank():
    """The same worker with a single rank (world size 1 over RCCL): what a one-GPU box CAN check — process-group set-up on
    "nccl", both gather modes, the comparison itself."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    out = _run(cmd, env)
    assert out.returncode == 0, out.stdout[-4000:]
    assert "DIST_GPU_OK 0 of 1" in out.stdout, out.stdout[-4000:]


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_on_one_gpu_over_gloo(world):
    """A one-GPU box's rehearsal of world > 1 (RCCL refuses two ranks on one device, so the transport is gloo): `world`
    processes, each with its own handle on the same GPU, every shard carrying its global env ids (ragged blocks: 6 * world + 1
    environments), both gather modes on device tensors, every gathered step compared with the single-rank run of the whole
    batch — bit for bit."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", ANTSRL_DIST_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    out = _run(cmd, env)
    assert out.returncode == 0, out.stdout[-6000:]
    assert all("DIST_GPU_OK %d of %d" % (r, world) in out.stdout for r in range(world)), out.stdout[-4000:]


def test_bench_py_with_two_ranks_on_one_gpu():
    """bench.py ITSELF under torch.distributed.run with two processes — the launch the driver makes on its 8-GPU node, which had
    never met a second process: per-rank set_device, per-rank tune_placement, shards with their global env ids, the
    per-step gather with its self-check, the MAX over ranks, the per-rank arrays of the JSON line.  One GPU here, so both ranks
    sit on device 0 and the transport is gloo (ANTSRL_BENCH_BACKEND / ANTSRL_BENCH_ONE_GPU); no scaling figure is read off it."""
    import json
    E, repeats = 96, 2   # (96 envs x 512 ants: the outputs are past vmm.SMALL_BYTES, so the placement tuner really runs)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", ANTSRL_BENCH_BACKEND="gloo", ANTSRL_BENCH_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--age", "30", "--repeats", str(repeats), "--envs", str(E)]
    out = _run(cmd, env)
    assert out.returncode == 0, out.stdout[-6000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, "rank 0 prints ONE JSON line:\n" + out.stdout[-3000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["collective_backend"] == "gloo"
    assert rec["env_id_base_per_rank"] == [0, E] and rec["n_envs_total"] == 2 * E
    assert rec["gather_self_checks"] == repeats and rec["scaling"] == "weak"
    for key in ("ms_per_step_per_rank", "device_unique_id_per_rank", "placement_chosen_ms_per_rank", "placement_default_ms_per_rank",
                "gather_overhead_us"):
        assert len(rec[key]) == 2, key
    assert all(v is not None and v > 0 for v in rec["ms_per_step_per_rank"] + rec["placement_chosen_ms_per_rank"])
    assert rec["device_unique_id_per_rank"][0] == rec["device_unique_id_per_rank"][1] is not None  # (one GPU)
    assert max(rec["ms_per_step_per_rank"]) <= rec["ms_per_step"] * 1.0001  # the line's figure is the MAX over the ranks
    assert rec["value"] == pytest.approx(2 * E * 512 * 20 / (rec["ms_per_step"] * 1e-3 * 20), rel=1e-6)
    log = os.path.join(ROOT, "gpurun_out", "bench_two_ranks_one_gpu.json")
    if os.path.isdir(os.path.dirname(log)):
        open(log, "w").write(lines[0] + "\n")
