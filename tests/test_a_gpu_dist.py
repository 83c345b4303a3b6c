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
