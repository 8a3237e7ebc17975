"""Rank program of tests/test_gpu_multirank.py (started by ``python -m torch.distributed.run``; not collected by pytest).

Every rank: the same seeded model, TrainingStep on ITS shard of a seeded global batch, then checks of the data-parallel
exchange (reference train.py:197) on the real HIP path:
  (a) the all-reduced flat gradient is identical on all ranks and equals the average of the per-shard gradients that
      rank 0 recomputes alone, shard by shard, on a second model (RCCL all-reduce of the real model.flat_gradients());
  (b) after clip + SGD every replica holds the same parameters, equal to a single-process step with that average;
  (c) a non-finite loss on ONE rank makes every rank skip the step (1-element MAX), parameters untouched;
  (d) sync_parameters: a rank that starts from different weights is overwritten by rank 0's.
Backend: nccl (= RCCL) by default, one GPU per rank; ENDO_DIST_BACKEND=gloo + ENDO_BENCH_SHARE_GPU=1 runs the same
program with all ranks on GPU 0 (plumbing check on a one-GPU box)."""
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")


def make_model(dev, seed):
    torch.manual_seed(seed)
    model = ea.FCDenseNet57(1)
    ea.utils.kaiming_weight_zero_bias(model, distribution="normal")
    with torch.no_grad():
        model.finalConv.bias.add_(8.0)          # depth away from zero (DepthScalingLayer divides by it)
    return model.to(dev).train()


def main():
    rank, world, local = ea.distributed.init_from_env()
    if os.environ.get("ENDO_BENCH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    n, h, w = 2, 64, 96
    batch = ea.synthetic.make_batch(n * world, h, w, seed=91, sparse_points=400)
    lo, hi = ea.distributed.shard_range(n * world, rank, world)
    mine = {k: v[lo:hi].contiguous().to(dev) for k, v in batch.items()}

    # (d) replicas that start apart are pulled onto rank 0's state
    model = make_model(dev, 100 + rank)
    opt = ea.optim.FusedClipSGD(model, lr=1.0e-3)
    step = ea.train_step.TrainingStep(model, opt, h, w)          # broadcasts parameters, BN statistics, momentum
    ref_model = make_model(dev, 100)
    assert torch.equal(model.flat_parameters(), ref_model.flat_parameters()), "sync_parameters did not take rank 0's weights"

    # (a) gradient exchange
    loss, _, _, _ = step.losses(mine)
    opt.zero_grad()
    loss.backward()
    scale = step.bucket.all_reduce()
    assert scale == 1.0 / world
    reduced = model.flat_gradients().clone() * scale
    gathered = [torch.zeros_like(reduced) for _ in range(world)]
    dist.all_gather(gathered, reduced)
    for other in gathered:
        assert torch.equal(other, gathered[0]), "ranks disagree on the reduced gradient"
    ref_opt = ea.optim.FusedClipSGD(ref_model, lr=1.0e-3)
    ref_step = ea.train_step.TrainingStep.__new__(ea.train_step.TrainingStep)          # same glue, no second broadcast
    ref_step.__dict__.update(step.__dict__)
    ref_step.model, ref_step.optimizer = ref_model, ref_opt
    total = torch.zeros_like(reduced)
    for r in range(world):
        a, b = ea.distributed.shard_range(n * world, r, world)
        shard = {k: v[a:b].contiguous().to(dev) for k, v in batch.items()}
        ref_opt.zero_grad()
        l, _, _, _ = ref_step.losses(shard)
        l.backward()
        total += ref_model.flat_gradients()
    want = total / world
    err = float((reduced - want).abs().max()) / float(want.abs().max())
    assert err <= 2e-5, "reduced gradient differs from the average of the shard gradients: %.3e" % err

    # (b) the step itself
    norm = opt.step(grad_scale=scale)
    ref_model.flat_gradients().copy_(want)
    ref_norm = ref_opt.step(grad_scale=1.0)
    assert abs(float(norm) - float(ref_norm)) <= 1e-5 * float(ref_norm)
    params = [torch.zeros_like(model.flat_parameters()) for _ in range(world)]
    dist.all_gather(params, model.flat_parameters())
    for other in params:
        assert torch.equal(other, params[0]), "replicas diverged after the step"
    err = float((model.flat_parameters() - ref_model.flat_parameters()).abs().max())
    assert err <= 1e-6, "parameters after the step differ from the single-process step: %.3e" % err

    # (c) one rank sees a NaN: everybody skips
    before = model.flat_parameters().clone()
    bad = {k: v.clone() for k, v in mine.items()}
    if rank == world - 1:
        bad["sparse_flows_1"][0, 0, h // 2, w // 2] = float("nan")
        bad["sparse_flow_masks_1"][0, 0, h // 2, w // 2] = 1.0
        bad["boundaries"][0, 0, h // 2, w // 2] = 1.0
    out = step(bad, lr=1.0e-3)
    assert out["skipped"], "rank %d did not take the guarded branch" % rank
    assert torch.equal(before, model.flat_parameters())
    # and a clean step afterwards still runs on all ranks
    out = step(mine, lr=1.0e-3)
    assert not out["skipped"]
    dist.barrier()
    if rank == 0:
        print("MULTIRANK_OK world=%d backend=%s" % (world, dist.get_backend()))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
