"""World-size-2 check of the data-parallel exchange step on CPU (gloo): one summed all-reduce of the
flat gradient bucket, the 1/world factor handed to the optimizer, the non-finite-loss consensus,
and shard bookkeeping.  The GPU box runs the same code over RCCL."""

import importlib
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = ea.distributed.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and ea.distributed.world_size() == world
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    bucket = ea.distributed.GradientBucket(lambda: flat)
    scale = bucket.all_reduce()
    assert scale == 0.5
    assert torch.equal(flat, torch.arange(1000, dtype=torch.float32) * 3)
    flag = ea.distributed.agree_nonfinite(torch.tensor([1.0 if rank == 1 else 0.0]))
    assert float(flag) == 1.0                                    # every rank takes the guarded branch
    flag = ea.distributed.agree_nonfinite(torch.tensor([0.0]))
    assert float(flag) == 0.0
    bn = torch.full((8,), float(rank))
    ea.distributed.broadcast_buffers(bn, src=0)
    assert float(bn.sum()) == 0.0
    logged = ea.distributed.mean_scalars(torch.tensor([1.0 + rank, 2.0]))
    assert torch.allclose(logged, torch.tensor([1.5, 2.0]))
    lo, hi = ea.distributed.shard_range(16, rank, world)
    out[rank] = (lo, hi)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange():
    port = _free_port()
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: (0, 8), 1: (8, 16)}


# ---------------------------------------------------------------------------------------------
# the exchange step with a real network behind it (CPU: the oracle plays the replica)
# ---------------------------------------------------------------------------------------------
def _shard(batch, lo, hi):
    return {k: v[lo:hi].clone() for k, v in batch.items()}


def _replica_worker(rank, world, port, out):
    """Rank r of a 2-rank job: the oracle's forward/backward on ITS shard (own BatchNorm batch statistics, like a
    DataParallel replica -- reference train.py:197), ONE summed all-reduce of the flat gradient through GradientBucket,
    the 1/world factor, the agreed non-finite flag, clip + SGD on the replica's own parameters."""
    from oracle import network as onet, schedule as osch, train_step as ostep
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ea.distributed.init_from_env(backend="gloo")
    torch.set_num_threads(2)
    state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(71), 72), bias=8.0)
    batch = ea.synthetic.make_batch(2 * world, 32, 32, seed=73, sparse_points=120)
    lo, hi = ea.distributed.shard_range(2 * world, rank, world)
    res = ostep.forward_backward(state, _shard(batch, lo, hi))
    names = onet.trainable_names()
    flat = torch.cat([res["grads"][nm].reshape(-1) for nm in names])
    scale = ea.distributed.GradientBucket(lambda: flat).all_reduce()
    bad = ea.distributed.agree_nonfinite(torch.tensor([0.0 if torch.isfinite(res["loss"]) else 1.0]))
    assert float(bad) == 0.0
    grads, off = [], 0
    for nm in names:
        cnt = state[nm].numel()
        grads.append((flat[off:off + cnt] * scale).view(state[nm].shape).clone())
        off += cnt
    params = [state[nm] for nm in names]
    norm = osch.clip_and_sgd(params, grads, [None] * len(names), 1.0e-3)
    out[rank] = (float(res["loss"]), float(norm), torch.cat([p.reshape(-1) for p in params]).numpy(),
                 (flat * scale).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_two_replicas_on_one_process():
    """2 ranks x N/2 samples == one process that runs the two shards as two replicas and averages their gradients
    (nn.DataParallel semantics: per-replica BatchNorm statistics, loss = mean of the shard means -- SURVEY.md 8(e)):
    identical averaged gradient on both ranks, equal to the single-process average to fp32 summation order; identical
    parameters after clip + SGD on every rank."""
    from oracle import network as onet, schedule as osch, train_step as ostep
    port = _free_port()
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_replica_worker, args=(2, port, out), nprocs=2, join=True)
    torch.set_num_threads(2)
    names = onet.trainable_names()
    batch = ea.synthetic.make_batch(4, 32, 32, seed=73, sparse_points=120)
    total, losses = None, []
    for lo, hi in ((0, 2), (2, 4)):
        state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(71), 72), bias=8.0)
        res = ostep.forward_backward(state, _shard(batch, lo, hi))
        flat = torch.cat([res["grads"][nm].reshape(-1) for nm in names])
        total = flat if total is None else total + flat
        losses.append(float(res["loss"]))
    mean_grad = 0.5 * total
    state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(71), 72), bias=8.0)
    grads, off = [], 0
    for nm in names:
        cnt = state[nm].numel()
        grads.append(mean_grad[off:off + cnt].view(state[nm].shape).clone())
        off += cnt
    params = [state[nm].clone() for nm in names]
    norm = osch.clip_and_sgd(params, grads, [None] * len(names), 1.0e-3)
    want_params = torch.cat([p.reshape(-1) for p in params]).numpy()
    r0, r1 = out[0], out[1]
    assert abs(r0[0] - losses[0]) < 1e-6 and abs(r1[0] - losses[1]) < 1e-6          # each rank saw its own shard
    assert (r0[3] == r1[3]).all(), "ranks disagree on the reduced gradient"
    assert (r0[2] == r1[2]).all(), "replicas diverged after the step"
    scale = float(mean_grad.abs().max())
    assert float(abs(torch.from_numpy(r0[3]) - mean_grad).max()) <= 1e-6 * scale
    assert abs(r0[1] - float(norm)) <= 1e-5 * float(norm)
    assert float(abs(torch.from_numpy(r0[2] - want_params)).max()) <= 1e-7


def test_bench_refuses_to_shrink_the_job():
    """`bench.py --gpus 2` where two GPUs are not visible must fail without printing a result line (it used to run one
    rank and report n_gpus = 1)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ENDO_BENCH_SHARE_GPU")}
    env["HIP_VISIBLE_DEVICES"] = env.get("HIP_VISIBLE_DEVICES", "")
    if torch.cuda.device_count() >= 2:
        env["HIP_VISIBLE_DEVICES"] = "0"
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert proc.returncode != 0
    assert "metric" not in proc.stdout
    assert "refusing" in proc.stderr
