"""Data parallelism: one process per GPU, persistent replicas, ONE all-reduce of the flat gradient
bucket per step (RCCL over xGMI via torch.distributed backend "nccl"), replacing the per-forward
broadcast / scatter / gather / reduce_add of ``torch.nn.DataParallel`` (reference train.py:197;
SURVEY.md 2.2 rows C1-C3).  The bucket's trailing float carries the non-finite-loss flag, so the same
collective makes every rank take the same branch of the guard (train.py:317-322).  Backend-agnostic: the CPU tests drive it
with gloo.
"""

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun).

    backend: "nccl" (= RCCL on ROCm) or "gloo"; default from ENDO_DIST_BACKEND, else nccl when a GPU is visible.  The
    environment defaults are set before the first HIP call of this function (the runtime reads them when it
    initialises; ``torch.cuda.device_count()`` does not initialise it, ``is_available()`` / ``set_device`` do)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("ENDO_DIST_BACKEND") or ("nccl" if torch.cuda.device_count() > 0 else "gloo")
        if backend == "nccl":          # RCCL wants one device per rank, selected before the communicator is built
            if local >= torch.cuda.device_count():
                raise RuntimeError("LOCAL_RANK %d but only %d GPU(s) are visible" % (local, torch.cuda.device_count()))
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_range(total, rank, world):
    """Samples [lo, hi) of a global batch owned by ``rank`` (equal shards)."""
    if total % world != 0:
        raise ValueError("global batch %d is not divisible by world size %d" % (total, world))
    per = total // world
    return rank * per, (rank + 1) * per


class GradientBucket(object):
    """The flat fp32 gradient buffer of a model replica, reduced with a single collective.  ``bucket_fn`` (optional) returns the
    buffer WITH its trailing guard slot (``FCDenseNet.flat_gradient_bucket``): the step's non-finite-loss flag then rides in the same
    all-reduce -- its sum over ranks is non-zero exactly when some rank's loss was NaN / Inf, so every rank takes the same branch of
    train.py:317-322 without a second collective and without the host in the loop."""

    def __init__(self, flat_grad_fn, bucket_fn=None):
        self._flat_grad_fn = flat_grad_fn
        self._bucket_fn = bucket_fn

    def all_reduce(self, flag=None):
        """Sum the bucket over ranks.  Returns the factor (1/world) the optimizer must apply -- it is folded into the fused clip+SGD
        kernel instead of costing a pass of its own -- or, when ``flag`` (1-element fp32 device tensor, this rank's guard flag) is
        given, ``(factor, flag after consensus)``: a 1-element view the optimizer kernel reads."""
        world = world_size()
        if flag is None:
            if world > 1:
                dist.all_reduce(self._flat_grad_fn(), op=dist.ReduceOp.SUM)
            return 1.0 / world
        if world <= 1:
            return 1.0, flag
        if self._bucket_fn is None:
            raise RuntimeError("GradientBucket: the guard flag needs the bucket with its trailing slot (bucket_fn)")
        bucket = self._bucket_fn()
        bucket[-1:].copy_(flag.reshape(1))
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
        return 1.0 / world, bucket[-1:]


def agree_nonfinite(flag_tensor):
    """flag_tensor: 1-element tensor, 1 where this rank's loss is NaN/Inf.  MAX over ranks so the
    guard of train.py:317-322 is taken by all ranks or none.  (TrainingStep no longer calls it: the flag travels in the gradient
    bucket, GradientBucket.all_reduce(flag); kept for callers that drive the modules themselves.)"""
    if world_size() > 1:
        dist.all_reduce(flag_tensor, op=dist.ReduceOp.MAX)
    return flag_tensor


def broadcast_buffers(flat_bn, src=0):
    """BN running statistics are per replica (DataParallel semantics: replica 0's survive).  Call
    before checkpointing so every rank saves rank 0's statistics."""
    if world_size() > 1:
        dist.broadcast(flat_bn, src=src)
    return flat_bn


def sync_parameters(model, optimizer=None, src=0):
    """Make every replica start from rank ``src``'s state: parameters, BN running statistics and (when the optimizer
    already has one) the momentum buffer.  ``nn.DataParallel`` re-broadcasts the module on every forward (reference
    train.py:197); persistent replicas need it once -- after construction, and after any rank-local change such as
    loading a checkpoint on one rank.  TrainingStep calls it when the world has more than one rank."""
    if world_size() <= 1:
        return
    dist.broadcast(model.flat_parameters(), src=src)
    flat_bn = getattr(model, "_flat_bn", None)
    if flat_bn is not None:
        dist.broadcast(flat_bn, src=src)
    nbt = getattr(model, "_nbt", None)
    if nbt is not None:
        dist.broadcast(nbt, src=src)
    momentum = getattr(optimizer, "_momentum", None) if optimizer is not None else None
    has = torch.tensor([1 if momentum is not None else 0], device=model.flat_parameters().device)
    dist.all_reduce(has, op=dist.ReduceOp.MAX)
    if int(has) > 0:
        if momentum is None:
            optimizer._ensure_state()
            momentum = optimizer._momentum
        dist.broadcast(momentum, src=src)
        steps = torch.tensor([optimizer._steps], device=momentum.device)
        dist.broadcast(steps, src=src)
        optimizer._steps = int(steps)


def mean_scalars(values):
    """all-reduce(mean) of a small tensor of logged scalars (loss terms)."""
    world = world_size()
    if world > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM)
        values /= world
    return values
