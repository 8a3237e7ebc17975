#!/usr/bin/env python3
"""Headline benchmark: training frame-pairs/s of the full hot-path step (BASELINE.json metric) on
synthetic 256 x 320 frame pairs, batch 8 per GPU, fp32 -- config 2 of BASELINE.json.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = reference train.py:272-328 on the MI355X path: boundary masking, two FC-DenseNet57
forwards, depth scaling, flow-from-depth, sparse-flow loss, depth warping, depth-consistency loss,
loss.item(), backward, one gradient all-reduce (N > 1), fused clip_grad_norm_(10) + SGD(0.9).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      the dominant kernel family (dense-layer conv3x3 forward/dgrad/wgrad: whichever took the most
                time in the warm-up steps, where all three carry HIP events), algorithmic FLOPs / HIP-event time
                of its launches measured live over the timed steps, on the launch stream; traffic = HBM bytes per
                launch from the rocprofv3 PMC passes committed under profiles/ (tools/pmc_traffic.py)
  cpu_baseline  the CPU oracle (oracle/, a port of the reference path) timed on this host's cores on
                a bounded sample (batch-1 steps)
"""

import argparse
import ctypes
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HEIGHT, WIDTH, BATCH = 256, 320, 8
FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix
HBM_PEAK_GBS = 8000.0
PAIR_GFLOP = 192.752                   # algorithmic conv work per frame pair (fwd+dgrad+wgrad)
MFMA_FAMILIES = (0, 5, 6)              # conv3x3_dense_fwd, dgrad_dense, wgrad_dense (endo_hip.h prof families)


def prof_read(lib, family):
    ms, cnt, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    rc = lib.endo_prof_read(family, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by))
    if rc != 0:
        raise RuntimeError("endo_prof_read failed: %d" % rc)
    return ms.value, cnt.value, fl.value, by.value


def cpu_baseline(seconds_budget=25.0):
    """The oracle's full training iteration on the host CPU, batch 1 at 256 x 320 (bounded sample)."""
    from oracle import network as onet, train_step as ostep      # checker / timed baseline only
    pkg = importlib.import_module("endoscopydepthestimation-pytorch_amd")
    # oneDNN at batch 1 stops scaling (and oversubscribes badly on a shared 256-thread host) well
    # before the full core count; 32 threads is what we actually use and report
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    state = onet.synthetic_state(10085)
    momentum = {}
    batch = pkg.synthetic.make_batch(1, HEIGHT, WIDTH, seed=0)
    ostep.train_iteration(state, momentum, batch, 1.0e-3)            # warm-up
    times = []
    start = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - start < seconds_budget and len(times) < 6):
        t0 = time.perf_counter()
        ostep.train_iteration(state, momentum, batch, 1.0e-3)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - start > 2.5 * seconds_budget:
            break
    times.sort()
    median = times[len(times) // 2]
    return {"value": 1.0 / median, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
            "sample": "%d full training iterations of the CPU oracle at batch 1, 256x320 (median; the GPU workload is batch 8)" % len(times),
            "ms_per_step": median * 1e3}


def pmc_traffic(family):
    """HBM bytes per step of `family`, from the newest committed rocprofv3 --pmc summary under profiles/ (a PMC pass
    serialises the kernels, so it cannot be taken inside the timed run; tools/pmc_traffic.py makes the file from the same
    bench.py command).  None when there is no such file."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_traffic.json")))
    for path in reversed(files):
        try:
            with open(path) as fh:
                entry = json.load(fh)["families"].get(family)
        except (OSError, ValueError, KeyError):
            continue
        if entry:
            return entry["traffic_bytes_per_step"], "profiles/" + os.path.basename(path)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print a per-family time table to stderr")
    args = ap.parse_args()

    pkg = importlib.import_module("endoscopydepthestimation-pytorch_amd")
    rank, world, local = pkg.distributed.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = pkg._lib.load()

    torch.manual_seed(10085)                                            # reference train.py:80
    model = pkg.models.FCDenseNet57(n_classes=1)
    pkg.utils.kaiming_weight_zero_bias(model, mode="fan_in", activation_mode="relu", distribution="normal")
    model = model.to(dev).train()
    optimizer = pkg.optim.FusedClipSGD(model, lr=1.0e-3, momentum=0.9, max_norm=10.0)
    scheduler = pkg.scheduler.CyclicLR(optimizer, base_lr=1.0e-4, max_lr=1.0e-3, step_size=2000)
    step_fn = pkg.train_step.TrainingStep(model, optimizer, HEIGHT, WIDTH, sfl_weight=20.0, dcl_weight=0.1)
    batch = {k: v.to(dev) for k, v in pkg.synthetic.make_batch(BATCH, HEIGHT, WIDTH, seed=rank).items()}

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Warm-up: all three dense-layer conv families carry HIP events (around every launch, on the launch stream), with the
    # weight-gradient side stream switched OFF so that kernels run one at a time: that gives each family's stand-alone
    # duration ("roofline_serial") and names the dominant family.  The last warm-up step and the timed steps run the
    # product configuration -- weight gradients overlapped with the data-gradient chain -- with events only on the
    # dominant family (two events per launch serialise neighbouring kernels; 88 launches of the other two families need
    # not pay for it).  Under overlap a kernel shares the chip with its neighbour, so its duration in the timed region is
    # longer than stand-alone: `roofline` reports what the timed region measured, `roofline_serial` the stand-alone figure.
    all_mask = 0
    for f in MFMA_FAMILIES:
        all_mask |= 1 << f
    it = 0
    serial_steps = max(args.warmup - 1, 0)
    lib.endo_set_wgrad_overlap(0)
    lib.endo_prof_enable(all_mask if serial_steps > 0 else 0)
    for _ in range(serial_steps):
        scheduler.batch_step(batch_iteration=it)
        step_fn(batch)
        it += 1
    barrier()
    fam_warm = {f: prof_read(lib, f) for f in MFMA_FAMILIES} if serial_steps > 0 else None
    lib.endo_prof_enable(0)
    lib.endo_set_wgrad_overlap(1)
    for _ in range(args.warmup - serial_steps):
        scheduler.batch_step(batch_iteration=it)
        step_fn(batch)
        it += 1
    barrier()
    if fam_warm is not None:
        dominant = max(MFMA_FAMILIES, key=lambda f: fam_warm[f][0])
        mask = 1 << dominant
    else:
        dominant, mask = None, all_mask
    lib.endo_prof_enable(mask)
    t0 = time.perf_counter()
    skipped = 0
    for _ in range(args.steps):
        scheduler.batch_step(batch_iteration=it)
        out = step_fn(batch)
        skipped += int(out["skipped"])
        it += 1
    barrier()
    elapsed = time.perf_counter() - t0
    fam = {f: prof_read(lib, f) for f in MFMA_FAMILIES}
    lib.endo_prof_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)

    # second metric of BASELINE.json: depth-warp (+ consistency loss) fwd+bwd, both directions
    warp = pkg.models.DepthWarpingLayer()
    dcl = pkg.losses.NormalizedDistanceLoss(HEIGHT, WIDTH)
    d1 = pkg.synthetic.smooth_depth(BATCH, HEIGHT, WIDTH, seed=1).to(dev).requires_grad_(True)
    d2 = pkg.synthetic.smooth_depth(BATCH, HEIGHT, WIDTH, seed=2).to(dev).requires_grad_(True)

    def warp_both():
        w21, i1 = warp([d1, d2, batch["boundaries"], batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"], batch["intrinsics"]])
        w12, i2 = warp([d2, d1, batch["boundaries"], batch["translations_2_wrt_1"], batch["rotations_2_wrt_1"], batch["intrinsics"]])
        loss = dcl([d1, w21, i1, batch["intrinsics"]]) + dcl([d2, w12, i2, batch["intrinsics"]])
        d1.grad = d2.grad = None
        loss.backward()

    for _ in range(3):
        warp_both()
    torch.cuda.synchronize()
    tw = time.perf_counter()
    reps = 20
    for _ in range(reps):
        warp_both()
    torch.cuda.synchronize()
    warp_ms_per_pair = (time.perf_counter() - tw) / reps / BATCH * 1e3

    breakdown = None
    if args.breakdown and rank == 0:
        lib.endo_prof_enable(-1)
        scheduler.batch_step(batch_iteration=it)
        step_fn(batch)
        torch.cuda.synchronize()
        breakdown = {}
        for f in range(16):
            ms, cnt, fl, by = prof_read(lib, f)
            if cnt:
                breakdown[lib.endo_prof_family_name(f).decode()] = {"ms": round(ms, 3), "launches": cnt,
                                                                    "tflops": round(fl / ms / 1e9, 2) if ms > 0 else None,
                                                                    "gbs": round(by / ms / 1e6, 1) if ms > 0 else None}
        lib.endo_prof_enable(0)
        print(json.dumps({"family_breakdown_one_step": breakdown}), file=sys.stderr)

    if rank != 0:
        return
    pairs = BATCH * world * args.steps
    if dominant is None:
        dominant = max(MFMA_FAMILIES, key=lambda f: fam[f][0])
    ms, cnt, fl, by = fam[dominant]
    achieved = fl / ms / 1e9 if ms > 0 else 0.0                         # TFLOP/s
    dom_name = lib.endo_prof_family_name(dominant).decode()
    traffic, traffic_src = pmc_traffic(dom_name)          # bytes per step -> per launch with the launches counted here
    if traffic is not None and cnt:
        traffic = traffic / (cnt / args.steps)
    result = {
        "metric": "train frame-pairs/sec at 256x320 bs=8",
        "value": pairs / elapsed,
        "unit": "frame-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "full training step (FC-DenseNet57 x2 fwd+bwd, depth scaling, flow, warp, losses, clip+SGD), "
                               "256x320, batch 8 per GPU, fp32 (BASELINE.json configs[1])",
                   "global_batch": BATCH * world, "height": HEIGHT, "width": WIDTH, "parallelism": "dp%d" % world},
        "skipped_steps": skipped,
        "conv_roofline_frac_whole_step": (pairs / elapsed) * PAIR_GFLOP / 1e3 / (FP32_MFMA_PEAK_TFLOPS * world),
        "depth_warp_fwd_bwd_ms_per_pair": warp_ms_per_pair,
        "roofline": {"kernel": dom_name, "bound": "mfma", "achieved": achieved,
                     "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                     "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC 2*FETCH_SIZE+WRITE_SIZE)",
                     "traffic_source": traffic_src, "algorithmic_bytes_per_launch": by / cnt if cnt else None,
                     "launches": cnt, "avg_launch_ms": ms / cnt if cnt else None,
                     "algorithmic_gbs": by / ms / 1e6 if ms > 0 else None,
                     "concurrent": True},
        "roofline_serial": None if fam_warm is None else {
            "note": "stand-alone kernel durations: warm-up steps with the weight-gradient side stream off (endo_set_wgrad_overlap(0))",
            "kernel": dom_name, "achieved": fam_warm[dominant][2] / fam_warm[dominant][0] / 1e9,
            "frac": fam_warm[dominant][2] / fam_warm[dominant][0] / 1e9 / FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "families_ms_per_step": {lib.endo_prof_family_name(f).decode(): fam_warm[f][0] / serial_steps for f in MFMA_FAMILIES}},
    }
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    print(json.dumps(result))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
