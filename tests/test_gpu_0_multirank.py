"""Data-parallel exchange on the GPU: N fresh rank processes (``python -m torch.distributed.run``) running
tests/multirank_worker.py.  The file sorts before the other GPU tests on purpose: the rank processes are started while
this pytest process has not yet initialised the GPU.

  * RCCL, one GPU per rank: needs >= 2 visible GPUs, skipped otherwise.
  * gloo with both ranks on GPU 0: the same program as a plumbing check on a one-GPU box (everything but RCCL itself:
    rendezvous, parameter broadcast, bucket all-reduce of the real gradient, consensus on the non-finite guard)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(key, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "multirank_worker.py")]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert proc.returncode == 0 and "MULTIRANK_OK world=%d" % world in proc.stdout, proc.stdout[-4000:]
    return proc.stdout


def test_two_ranks_rccl():
    if torch.cuda.device_count() < 2:          # device_count() does not initialise the GPU
        pytest.skip("needs two GPUs")
    out = _run(2, {})
    assert "backend=nccl" in out


def test_two_ranks_one_gpu_gloo():
    assert torch.cuda.device_count() >= 1, "GPU tests need an MI355X"
    out = _run(2, {"ENDO_DIST_BACKEND": "gloo", "ENDO_BENCH_SHARE_GPU": "1"})
    assert "backend=gloo" in out


def test_bench_two_ranks_on_one_gpu_is_marked_as_plumbing():
    """`bench.py --gpus 2` with ENDO_BENCH_SHARE_GPU=1 (gloo, both ranks on GPU 0): the launcher, the rendezvous, the flag-carrying
    bucket all-reduce inside the timed steps and rank 0's JSON line -- which must say what it is: two ranks, and not a measurement.
    (The first box with two GPUs runs test_two_ranks_rccl and `bench.py --gpus 2` over RCCL; no scaling number exists until then.)"""
    import json
    assert torch.cuda.device_count() >= 1, "GPU tests need an MI355X"
    env = dict(os.environ)
    env.update({"ENDO_DIST_BACKEND": "gloo", "ENDO_BENCH_SHARE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(key, None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2", "--no-cpu-baseline"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-4000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, proc.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["world"] == 2 and line["steps"] == 2
    assert line["shared_gpu_plumbing_check_not_a_measurement"] is True
    assert line["skipped_steps"] == 0 and line["value"] > 0
