"""World-size-2 / 4 / 8 checks of the data-parallel exchange step on CPU (gloo): one summed all-reduce of the
flat gradient bucket with the non-finite-loss flag in its trailing slot, the 1/world factor handed to the
optimizer, shard bookkeeping, sync_parameters after a rank-local load; and two / four oracle replicas against
one process.  The GPU box runs the same code over RCCL."""

import importlib
import os
import socket

import pytest
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
    torch.set_num_threads(1)
    assert (r, w) == (rank, world) and ea.distributed.world_size() == world
    tri = world * (world + 1) // 2
    flat = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    bucket = ea.distributed.GradientBucket(lambda: flat)
    scale = bucket.all_reduce()
    assert scale == 1.0 / world
    assert torch.equal(flat, torch.arange(1000, dtype=torch.float32) * tri)
    # the guard flag rides in the bucket's trailing slot (TrainingStep): ONE collective gives the summed gradients AND the consensus
    full = torch.cat([torch.arange(1000, dtype=torch.float32) * (rank + 1), torch.tensor([123.0])])          # stale slot content
    bucket = ea.distributed.GradientBucket(lambda: full[:1000], lambda: full)
    scale, flag = bucket.all_reduce(torch.tensor([1.0 if rank == world - 1 else 0.0]))
    assert scale == 1.0 / world and float(flag) == 1.0            # one rank's NaN: every rank sees a non-zero flag
    assert torch.equal(full[:1000], torch.arange(1000, dtype=torch.float32) * tri)
    scale, flag = bucket.all_reduce(torch.tensor([0.0]))
    assert float(flag) == 0.0
    scale, flag = bucket.all_reduce(torch.tensor([1.0]))
    assert float(flag) == float(world)                            # all ranks bad: still just "non-zero"
    flag = ea.distributed.agree_nonfinite(torch.tensor([1.0 if rank == 1 else 0.0]))
    assert float(flag) == 1.0                                    # (the stand-alone MAX consensus, for callers that drive the modules)
    flag = ea.distributed.agree_nonfinite(torch.tensor([0.0]))
    assert float(flag) == 0.0
    bn = torch.full((8,), float(rank))
    ea.distributed.broadcast_buffers(bn, src=0)
    assert float(bn.sum()) == 0.0
    logged = ea.distributed.mean_scalars(torch.tensor([1.0 + rank, 2.0]))
    assert torch.allclose(logged, torch.tensor([(world + 1) / 2.0, 2.0]))
    lo, hi = ea.distributed.shard_range(16, rank, world)
    # sync_parameters after a rank-local change (a checkpoint loaded on one rank): parameters, BN statistics, momentum and the
    # optimizer's step count of every replica become rank 0's
    model = _HostReplica(rank)
    opt = _HostOptimizer(rank)
    ea.distributed.sync_parameters(model, opt, src=0)
    assert float(model.flat_parameters().sum()) == 0.0 and float(model._flat_bn.sum()) == 0.0 and int(model._nbt) == 0
    assert opt._momentum is not None and float(opt._momentum.sum()) == 0.0 and opt._steps == 7
    out[rank] = (lo, hi)
    dist.barrier()
    dist.destroy_process_group()


class _HostReplica(object):
    """the attributes distributed.sync_parameters reads off FCDenseNet, on the CPU"""

    def __init__(self, rank):
        self._flat = torch.full((64,), float(rank))
        self._flat_bn = torch.full((16,), float(rank))
        self._nbt = torch.tensor(rank)

    def flat_parameters(self):
        return self._flat


class _HostOptimizer(object):
    """rank 0 has optimizer state (as after loading a checkpoint there), the others none yet"""

    def __init__(self, rank):
        self._momentum = torch.zeros(64) if rank == 0 else None
        self._steps = 7 if rank == 0 else 0

    def _ensure_state(self):
        self._momentum = torch.full((64,), 5.0)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gradient_exchange(world):
    port = _free_port()
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    per = 16 // world
    assert dict(out) == {r: (r * per, (r + 1) * per) for r in range(world)}


# ---------------------------------------------------------------------------------------------
# the exchange step with a real network behind it (CPU: the oracle plays the replica)
# ---------------------------------------------------------------------------------------------
def _shard(batch, lo, hi):
    return {k: v[lo:hi].clone() for k, v in batch.items()}


def _replica_worker(rank, world, port, out):
    """Rank r of a `world`-rank job: the oracle's forward/backward on ITS shard (own BatchNorm batch statistics, like a
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
    full = torch.cat([flat, torch.zeros(1)])
    flat = full[:-1]
    scale, bad = ea.distributed.GradientBucket(lambda: flat, lambda: full).all_reduce(torch.tensor([0.0 if torch.isfinite(res["loss"]) else 1.0]))
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


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_equal_replicas_on_one_process(world):
    """`world` ranks x 2 samples == one process that runs the shards as `world` replicas and averages their gradients
    (nn.DataParallel semantics: per-replica BatchNorm statistics, loss = mean of the shard means -- SURVEY.md 8(e)):
    identical averaged gradient on all ranks, equal to the single-process average to fp32 summation order; identical
    parameters after clip + SGD on every rank."""
    from oracle import network as onet, schedule as osch, train_step as ostep
    port = _free_port()
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_replica_worker, args=(world, port, out), nprocs=world, join=True)
    torch.set_num_threads(2)
    names = onet.trainable_names()
    batch = ea.synthetic.make_batch(2 * world, 32, 32, seed=73, sparse_points=120)
    total, losses = None, []
    for r in range(world):
        state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(71), 72), bias=8.0)
        res = ostep.forward_backward(state, _shard(batch, 2 * r, 2 * r + 2))
        flat = torch.cat([res["grads"][nm].reshape(-1) for nm in names])
        total = flat if total is None else total + flat
        losses.append(float(res["loss"]))
    mean_grad = total / world
    state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(71), 72), bias=8.0)
    grads, off = [], 0
    for nm in names:
        cnt = state[nm].numel()
        grads.append(mean_grad[off:off + cnt].view(state[nm].shape).clone())
        off += cnt
    params = [state[nm].clone() for nm in names]
    norm = osch.clip_and_sgd(params, grads, [None] * len(names), 1.0e-3)
    want_params = torch.cat([p.reshape(-1) for p in params]).numpy()
    r0 = out[0]
    for r in range(world):
        assert abs(out[r][0] - losses[r]) < 1e-6                                      # each rank saw its own shard
        assert (out[r][3] == r0[3]).all(), "ranks disagree on the reduced gradient"
        assert (out[r][2] == r0[2]).all(), "replicas diverged after the step"
    scale = float(mean_grad.abs().max())
    assert float(abs(torch.from_numpy(r0[3]) - mean_grad).max()) <= 2e-6 * scale          # another summation order than the ring's
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
