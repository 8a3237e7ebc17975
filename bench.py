#!/usr/bin/env python3
"""Headline benchmark: training frame-pairs/s of the full hot-path step (BASELINE.json metric) on
synthetic frame pairs -- by default config 2 of BASELINE.json (configs[1]): 256 x 320, batch 8 per GPU, fp32.

    python bench.py --gpus N --steps K --warmup W [--config 1|2|3|4|5|6]

With N > 1 and no WORLD_SIZE in the environment the script starts N fresh rank processes itself
(``python -m torch.distributed.run --nproc-per-node N ... bench.py``, before this process touches the GPU) and relays
rank 0's JSON line; under torchrun it is one of the ranks.  It never prints an ``n_gpus`` it did not reach.

A step = reference train.py:272-328 on the MI355X path: boundary masking, two FC-DenseNet57
forwards, depth scaling, flow-from-depth, sparse-flow loss, depth warping, depth-consistency loss,
loss.item(), backward, one gradient all-reduce (N > 1), fused clip_grad_norm_(10) + SGD(0.9).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

--config selects the BASELINE.json workload by its index in ``configs``:
    1   256 x 320, batch 8 per GPU, fp32                                   (default; the metric's configuration)
    3   512 x 640, batch 4 per GPU, fp32
    4   256 x 320, batch 8 per GPU, poses scaled by a per-sample frame gap U{5..30}/10 ("adjacent range 5-30"), FP16 STORAGE
        (the 16-bit-storage family compiled for IEEE half): the per-GPU half of BASELINE.json configs[4]; its own line
    6   (not a BASELINE config) the pose regime of configs[4] in the fp32 family (round 2's --config 4)
    2   256 x 320, batch 8 per GPU, BF16 STORAGE: the network over bf16 level buffers (endo_net16_fwd / endo_net16_bwd) -- the
        per-GPU half of BASELINE.json configs[2] (bs 64 bf16 over 8 GPUs); its own line, never compared with configs[1]
    5   (not a BASELINE config) 256 x 320, batch 8 per GPU, bf16 MFMA operands over fp32 tensors: the mixed-precision mode of the
        fp32 family (round 2's --config 2)

Extra objects on that line:
  roofline             the dominant kernel family (dense-layer conv3x3 forward/dgrad/wgrad: whichever took the most
                       time in the warm-up steps, where all three carry HIP events), algorithmic FLOPs / HIP-event time
                       of its launches measured live over the timed steps, on the launch stream; traffic = HBM bytes per
                       launch from the rocprofv3 PMC passes committed under profiles/ (tools/pmc_traffic.py)
  roofline_depth_warp  the second BASELINE metric (depth-warp fwd+bwd, both directions, + the consistency loss that
                       consumes it): HBM bound; algorithmic bytes / HIP-event kernel time, and the host-inclusive ms/pair
  cpu_baseline         the CPU oracle (oracle/, a port of the reference path) timed on this host's physical cores on a
                       bounded sample of the same workload (the GPU batch size) and at batch 1
"""

import argparse
import ctypes
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 matrix
HBM_PEAK_GBS = 8000.0
PAIR_GFLOP_256x320 = 192.752           # algorithmic conv work per frame pair (fwd+dgrad+wgrad)
MFMA_FAMILIES = (0, 5, 6)              # conv3x3_dense_fwd, dgrad_dense, wgrad_dense (endo_hip.h prof families)
FAMILY_GEOMETRY, FAMILY_LOSS = 10, 11

CONFIGS = {
    1: dict(height=256, width=320, batch=8, gap=None,
            metric="train frame-pairs/sec at 256x320 bs=8",
            workload="full training step (FC-DenseNet57 x2 fwd+bwd, depth scaling, flow, warp, losses, clip+SGD), "
                     "256x320, batch 8 per GPU, fp32 (BASELINE.json configs[1])"),
    5: dict(height=256, width=320, batch=8, gap=None, bf16_operands=True,
            metric="train frame-pairs/sec at 256x320 bs=8, bf16 MFMA operands",
            workload="full training step, 256x320, batch 8 per GPU, MIXED PRECISION: the dense layers' forward / data-gradient / "
                     "weight-gradient kernels round their MFMA operands to bf16 (fp32 accumulation; tensors in memory, BN, reductions, "
                     "geometry, losses and optimizer fp32) -- a mode of the fp32 family, not a BASELINE.json config (the bf16 config is "
                     "--config 2); NOT comparable with the fp32 line of configs[1]"),
    3: dict(height=512, width=640, batch=4, gap=None,
            metric="train frame-pairs/sec at 512x640 bs=4",
            workload="full training step (FC-DenseNet57 x2 fwd+bwd, depth scaling, flow, warp, losses, clip+SGD), "
                     "512x640, batch 4 per GPU, fp32 (BASELINE.json configs[3])"),
    4: dict(height=256, width=320, batch=8, gap=(5, 30), fp16_storage=True,
            metric="train frame-pairs/sec at 256x320 bs=8, adjacent range 5-30, fp16 storage",
            workload="full training step, 256x320, batch 8 per GPU, poses scaled by a per-sample frame gap U{5..30}/10, FP16 STORAGE: the "
                     "network's activations and inter-layer gradients are stored as IEEE half in 32-channel blocks (endo_net16h_*: the "
                     "bf16-storage family compiled for half, power-of-two gradient scale per backward call), fp32 accumulation, statistics, "
                     "parameter gradients, geometry, losses, clipping and SGD -- the per-GPU half of BASELINE.json configs[4]; its own line"),
    6: dict(height=256, width=320, batch=8, gap=(5, 30),
            metric="train frame-pairs/sec at 256x320 bs=8, adjacent range 5-30, fp32",
            workload="full training step, 256x320, batch 8 per GPU, poses scaled by a per-sample frame gap U{5..30}/10, fp32 storage "
                     "(the pose regime of BASELINE.json configs[4] in the fp32 family; round 2's --config 4)"),
    2: dict(height=256, width=320, batch=8, gap=None, bf16_storage=True,
            metric="train frame-pairs/sec at 256x320 bs=8, bf16 storage",
            workload="full training step, 256x320, batch 8 per GPU, BF16 STORAGE: the network's activations and inter-layer gradients "
                     "are stored as bf16 in 32-channel blocks and multiplied on the bf16 matrix cores (fp32 accumulation; BatchNorm "
                     "statistics, parameter gradients, geometry, losses, clipping and SGD fp32) -- the per-GPU half of BASELINE.json "
                     "configs[2] (bs 64 bf16 over 8 GPUs); a different function from the fp32 line of configs[1], with its own parity "
                     "bounds (tests/test_gpu_bf16.py)"),
}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """--gpus N outside torchrun: N fresh rank processes (this process has not touched the GPU and never will).
    Relays the children's output; the exit code is theirs."""
    import torch
    have = torch.cuda.device_count()          # counts devices without initialising HIP on this image
    if have < args.gpus and not os.environ.get("ENDO_BENCH_SHARE_GPU"):
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) are visible; refusing to report a smaller job as %d GPUs\n"
                         % (args.gpus, have, args.gpus))
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    result = None
    for ln in lines:
        try:
            obj = json.loads(ln)
        except ValueError:
            sys.stderr.write(ln + "\n")
            continue
        if isinstance(obj, dict) and "metric" in obj:
            result = obj
    if proc.returncode != 0 or result is None:
        sys.stderr.write("bench.py: the %d-rank job failed (exit code %d)\n" % (args.gpus, proc.returncode))
        return proc.returncode or 1
    if result.get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: asked for %d ranks, the job reports %r\n" % (args.gpus, result.get("n_gpus")))
        return 1
    print(json.dumps(result))
    sys.stdout.flush()
    return 0


def prof_read(lib, family):
    ms, cnt, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    rc = lib.endo_prof_read(family, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by))
    if rc != 0:
        raise RuntimeError("endo_prof_read failed: %d" % rc)
    return ms.value, cnt.value, fl.value, by.value


def host_cpu():
    """(model string, physical cores this process may use, logical CPUs it may use)."""
    model = "unknown"
    cores = {}
    try:
        cpu = phys = core = None
        with open("/proc/cpuinfo") as fh:
            for line in fh.read().splitlines() + [""]:
                if not line.strip():
                    if cpu is not None:
                        cores[cpu] = (phys, core if core is not None else cpu)
                    cpu = phys = core = None
                    continue
                key, _, val = line.partition(":")
                key, val = key.strip(), val.strip()
                if key == "processor":
                    cpu = int(val)
                elif key == "model name":
                    model = val
                elif key == "physical id":
                    phys = int(val)
                elif key == "core id":
                    core = int(val)
    except (OSError, ValueError):
        pass
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    physical = len({cores[c] for c in allowed if c in cores}) or len(allowed)
    return model, physical, len(allowed)


def cpu_baseline(cfg):
    """The oracle's full training iteration on the host CPU, a bounded sample: batch 1 at two thread
    counts (all physical cores, and 32 -- oneDNN on a large shared host often runs slower on every core than on 32), then the
    GPU workload's batch at the better of the two.  `value` is the workload-batch figure; `cores` the threads it used."""
    import torch
    from oracle import network as onet, train_step as ostep      # checker / timed baseline only
    pkg = importlib.import_module("endoscopydepthestimation-pytorch_amd")
    model, physical, logical = host_cpu()
    h, w = cfg["height"], cfg["width"]

    def run(batch_size, threads, timed, give_up_after=None):
        torch.set_num_threads(threads)
        state = onet.synthetic_state(10085)
        momentum = {}
        batch = pkg.synthetic.make_batch(batch_size, h, w, seed=0, gap_scale=cfg["gap"])
        t0 = time.perf_counter()
        ostep.train_iteration(state, momentum, batch, 1.0e-3)            # warm-up (primitive creation, page faults)
        warm = time.perf_counter() - t0
        times = []
        if give_up_after is None or warm < give_up_after:
            for _ in range(timed):
                t0 = time.perf_counter()
                ostep.train_iteration(state, momentum, batch, 1.0e-3)
                times.append(time.perf_counter() - t0)
        times.sort()
        median = times[len(times) // 2] if times else warm
        return {"value": batch_size / median, "unit": "frame-pairs/s", "batch": batch_size, "threads": threads,
                "iterations": len(times), "ms_per_step": median * 1e3, "warmup_only": not times}

    counts = sorted({min(physical, 32), physical})
    sweep = [run(1, counts[0], 3)]
    for threads in counts[1:]:
        # a thread count whose warm-up alone takes 3x the best step so far is not going to win: one iteration is enough
        sweep.append(run(1, threads, 2, give_up_after=3.0 * sweep[0]["ms_per_step"] / 1e3))
    best = max(sweep, key=lambda r: r["value"])
    # median of >= 3 timed iterations after the warm-up (the 512x640 workload, ~4x the work per sample, gets 1 when a
    # batch-1 step already takes > 4 s: the whole baseline leg stays within ~2 minutes)
    full = run(cfg["batch"], best["threads"], 1 if best["ms_per_step"] > 4000.0 else 3)
    return {"value": full["value"], "unit": "frame-pairs/s", "cores": full["threads"], "kind": "port",
            "cpu_model": model, "physical_cores": physical, "logical_cpus": logical,
            "sample": "median of %d full training iteration(s) of the CPU oracle at batch %d, %dx%d (the GPU workload's batch) after 1 warm-up, "
                      "%d torch threads = the faster of {32, all %d physical cores} at batch 1" % (
                          max(full["iterations"], 1), cfg["batch"], h, w, full["threads"], physical),
            "ms_per_step": full["ms_per_step"], "timed_iterations": full["iterations"], "batch": cfg["batch"], "batch1_by_threads": sweep}


def _summary_of_this_tree(suffix):
    """The profiles/*<suffix> summary measured on the kernel sources this process runs (their sha256, tools/source_id.py): (doc, name),
    or (None, reason).  Summaries of other revisions stay in profiles/ as history, so the choice is by hash, not by file name."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_id import csrc_sha256
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)))
    if not files:
        return None, "no profiles/*" + suffix
    running = csrc_sha256()
    last_seen = None
    for path in reversed(files):
        name = "profiles/" + os.path.basename(path)
        try:
            with open(path) as fh:
                doc = json.load(fh)
        except (OSError, ValueError):
            continue
        measured_on = (doc.get("source") or {}).get("csrc_sha256")
        if measured_on == running:
            return doc, name
        if last_seen is None:
            last_seen = (name, measured_on)
    if last_seen is None:
        return None, "unreadable: profiles/*" + suffix
    return None, "stale: %s was measured on kernel sources %s, this tree is %s -- regenerate it with tools/final_profiles.sh" % (
        last_seen[0], (last_seen[1] or "of an unrecorded revision")[:12], running[:12])


def dispatches_per_step(config=1):
    """Kernel dispatches per training step, from the rocprofv3 kernel-trace summary under profiles/ that was taken on this tree's kernel sources (tools/summarize_rocprof.py
    writes `*_kernel_stats.json` next to the table) -- counted by the profiler, not estimated, and tied to the kernel sources by their
    hash: a summary of another revision is refused (None + the reason).  In-process counting was tried and dropped: this stack's
    CUDAGraph.debug_dump writes nothing and torch.profiler sees 175 of the ~600 dispatches (tests/diag/dispatch_count_probe.py)."""
    suffix = "_kernel_stats.json" if config in (1, 3, 6) else "_kernel_stats_config%d.json" % (2 if config == 4 else config)
    doc, name = _summary_of_this_tree(suffix)
    if doc is None:
        return {"value": None, "source": name}
    profiled = config in (1, 2)          # the trace is of configs[1] / configs[2]; the other configs launch the same kernels at other sizes
    return {"value": doc["dispatches_per_step"], "library_kernels": doc["library_kernel_dispatches_per_step"],
            "kernel_time_ms_per_step": doc["kernel_time_ms_per_step"] if profiled else None,
            "source": name if profiled else name + " (the same launches; that trace's sizes)"}


def pmc_traffic(family, config=1):
    """HBM bytes per step of `family`, from the committed rocprofv3 --pmc summary under profiles/ taken on this tree's kernel sources (a PMC pass
    serialises the kernels, so it cannot be taken inside the timed run; tools/pmc_traffic.py makes the file from the same
    bench.py command; `*_pmc_traffic.json` for configs[1], `*_pmc_traffic_config<k>.json` otherwise).  The file records the
    sha256 of the kernel sources it was measured on (tools/source_id.py): when that is not the tree this process runs, the
    traffic is reported as None with the reason instead of stale bytes.  Returns (bytes per step or None, source / reason)."""
    suffix = "_pmc_traffic.json" if config == 1 else "_pmc_traffic_config%d.json" % config
    doc, name = _summary_of_this_tree(suffix)
    if doc is None:
        return None, name
    entry = doc.get("families", {}).get(family)
    if not entry:
        return None, "%s has no family %s" % (name, family)
    return entry["traffic_bytes_per_step"], name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="index into BASELINE.json configs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print a per-family time table to stderr")
    ap.add_argument("--kernel-option", action="append", default=[], metavar="ID=VALUE",
                    help="development A/B: FCDenseNet57.set_kernel_option(ID, VALUE) (include/endo_hip.h ENDO_OPT_*); recorded in the line")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and env_world <= 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))          # before anything here touches the GPU
    if env_world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, env_world))

    import torch
    cfg = CONFIGS[args.config]
    height, width, batch_size = cfg["height"], cfg["width"], cfg["batch"]
    pair_gflop = PAIR_GFLOP_256x320 * (height * width) / (256.0 * 320.0)
    pkg = importlib.import_module("endoscopydepthestimation-pytorch_amd")
    rank, world, local = pkg.distributed.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # ENDO_BENCH_SHARE_GPU=1 (with ENDO_DIST_BACKEND=gloo): every rank on GPU 0 -- a plumbing check of the multi-rank path
    # on a one-GPU box (launcher, rendezvous, parameter broadcast, gradient all-reduce, non-finite consensus, rank-0 line);
    # the line it prints is marked and is not a measurement
    shared = bool(os.environ.get("ENDO_BENCH_SHARE_GPU")) and world > 1
    if shared:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib = pkg._lib.load()
    bf16 = bool(cfg.get("bf16_operands"))
    fp16_storage = bool(cfg.get("fp16_storage"))
    bf16_storage = bool(cfg.get("bf16_storage")) or fp16_storage          # the 16-bit-storage family (bf16 or half)

    torch.manual_seed(10085)                                            # reference train.py:80
    model = pkg.models.FCDenseNet57(n_classes=1)
    pkg.utils.kaiming_weight_zero_bias(model, mode="fan_in", activation_mode="relu", distribution="normal")
    model = model.to(dev).train()
    model.set_kernel_option(4, 1 if bf16 else 0)                       # ENDO_OPT_MFMA_BF16 (include/endo_hip.h): this model only
    for spec in args.kernel_option:
        option_id, value = (int(v) for v in spec.split("="))
        if option_id == 4:
            raise SystemExit("the operand precision belongs to --config, not to --kernel-option")
        model.set_kernel_option(option_id, value)
    optimizer = pkg.optim.FusedClipSGD(model, lr=1.0e-3, momentum=0.9, max_norm=10.0)
    scheduler = pkg.scheduler.CyclicLR(optimizer, base_lr=1.0e-4, max_lr=1.0e-3, step_size=2000)
    step_fn = pkg.train_step.TrainingStep(model, optimizer, height, width, sfl_weight=20.0, dcl_weight=0.1, bf16_storage=bf16_storage and not fp16_storage,
                                          fp16_storage=fp16_storage)
    batch = {k: v.to(dev) for k, v in pkg.synthetic.make_batch(batch_size, height, width, seed=rank, gap_scale=cfg["gap"]).items()}

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Warm-up (W = --warmup untimed steps in all): the LAST min(W, 3) of them run the product configuration -- weight gradients on the side
    # stream, overlapped with the data-gradient chain, events on nothing but a launch counter -- so that the timed region starts from the
    # steady state it measures (clocks, caches, the allocator's and the side stream's state).  The up to two steps before those carry HIP events
    # around every launch of the three dense-layer conv families (on the launch stream) with the side stream switched OFF, so that kernels run
    # one at a time: each family's stand-alone duration ("roofline_serial"), which also names the dominant family.  In the timed steps only
    # the dominant family carries events (two events per launch serialise neighbouring kernels).  Under overlap a kernel shares the chip with
    # its neighbour, so its duration in the timed region is longer than stand-alone: `roofline` reports what the timed region measured,
    # `roofline_serial` the stand-alone figure.
    all_mask = 0
    for f in MFMA_FAMILIES:
        all_mask |= 1 << f
    it = 0
    product_steps = min(args.warmup, 3)
    serial_steps = min(args.warmup - product_steps, 2)
    for _ in range(args.warmup - product_steps - serial_steps):          # --warmup > 5: the rest, uninstrumented, first
        scheduler.batch_step(batch_iteration=it)
        step_fn(batch)
        it += 1
    model.set_kernel_option(5, 0)                                      # ENDO_OPT_WGRAD_OVERLAP off: kernels one at a time
    model.set_wgrad_overlap16(False)                                   # the same for the 16-bit-storage family's handles
    lib.endo_prof_sample(1)
    lib.endo_prof_enable(all_mask if serial_steps > 0 else 0)
    for _ in range(serial_steps):
        scheduler.batch_step(batch_iteration=it)
        step_fn(batch)
        it += 1
    barrier()
    fam_warm = {f: prof_read(lib, f) for f in MFMA_FAMILIES} if serial_steps > 0 else None
    lib.endo_prof_enable(0)
    overlap_mode = 1          # ENDO_OPT_WGRAD_OVERLAP of the timed region: 1, or what --kernel-option 5=... asked for (2 = one fork per dense block)
    for spec in args.kernel_option:
        if int(spec.split("=")[0]) == 5:
            overlap_mode = int(spec.split("=")[1])
    model.set_kernel_option(5, overlap_mode)
    model.set_wgrad_overlap16(int(os.environ.get("ENDO16_WGRAD_OVERLAP", "1")))          # development A/B: 2 = one fork per dense block
    NEVER = 1 << 30                                                    # a sampling period nothing reaches: launches are counted, not timed
    lib.endo_prof_sample(NEVER)
    lib.endo_prof_enable(all_mask)                                     # (the first launch of each family gets one event pair; nothing else)
    for _ in range(product_steps):
        scheduler.batch_step(batch_iteration=it)
        step_fn(batch)
        it += 1
    barrier()
    launches_per_step = {}
    for f in MFMA_FAMILIES:
        n_seen = ctypes.c_int64(0)
        lib.endo_prof_seen(f, ctypes.byref(n_seen))
        launches_per_step[f] = n_seen.value // product_steps if product_steps else 0
    lib.endo_prof_enable(0)
    if fam_warm is not None:
        dominant = max(MFMA_FAMILIES, key=lambda f: fam_warm[f][0])
        mask = 1 << dominant
    else:
        dominant, mask = None, all_mask
    if os.environ.get("ENDO_BENCH_NO_EVENTS"):                         # development: what the per-launch events cost the timed region
        mask = 0
    # A dominant family on the caller's stream (the data gradient in fp32 since round 4, in the 16-bit modes always) is SAMPLED: the two
    # events around a launch serialise it with its neighbours -- timing all 44 launches of a step cost the 16-bit step 0.36 ms = 2.7 %
    # (profiles/r03_y_events_cost.txt), the fp32 step 0.5 % (409.0 against 411 frame-pairs/s).  One launch in `period` is timed, with
    # gcd(period, launches per step) = 1 (the launch count is the one just counted, not assumed), so `period` consecutive steps time every
    # launch of the step exactly once; only whole multiples of `period` steps are sampled (the remaining steps of the timed region run
    # untimed), so every launch enters the average equally often and launches_per_step x avg_launch_ms IS the family's time per step.
    # On the side stream (the fp32 weight gradients when they dominate) events cost nothing and every launch is timed.
    import math
    on_side_stream = dominant == 6 and not bf16_storage and overlap_mode != 0          # family 6 = wgrad_dense
    lps = launches_per_step.get(dominant, 0) if dominant is not None else 0
    period = 1
    if not on_side_stream and dominant is not None and lps > 1:
        period = next((q for q in (7, 5, 3, 11, 9, 13, 4, 2) if args.steps >= q and math.gcd(q, lps) == 1), 1)
    sampled_steps = (args.steps // period) * period
    lib.endo_prof_sample(period)
    lib.endo_prof_enable(mask)
    t0 = time.perf_counter()
    skipped = 0
    # every step's losses, guard flag and gradient norm are read on the host (the reference's loss.item(), train.py:317) -- one step
    # late: the guard itself is decided on the device (StepOutput, train_step.py), so the read of step k - 1 happens while step k is
    # queued and the GPU does not idle at the loss
    pending = None
    for k in range(args.steps):
        if k == sampled_steps and period > 1:
            lib.endo_prof_sample(NEVER)          # the even cover is complete: the rest of the timed region is counted, not timed
        scheduler.batch_step(batch_iteration=it)
        out = step_fn(batch)
        if pending is not None:
            skipped += int(pending["skipped"])
        pending = out
        it += 1
    skipped += int(pending["skipped"]) if pending is not None else 0
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0                             # this rank's own steps (before waiting for the others)
    barrier()
    elapsed = time.perf_counter() - t0
    fam = {f: prof_read(lib, f) for f in MFMA_FAMILIES}
    seen = {}
    for f in MFMA_FAMILIES:
        n_seen = ctypes.c_int64(0)
        lib.endo_prof_seen(f, ctypes.byref(n_seen))
        seen[f] = n_seen.value
    lib.endo_prof_enable(0)
    lib.endo_prof_sample(1)
    per_rank = [batch_size * args.steps / own_elapsed]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
        mine = torch.tensor([per_rank[0]], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(gathered, mine)
        per_rank = [float(g) for g in gathered]

    # second metric of BASELINE.json: depth-warp (+ consistency loss) fwd+bwd, both directions
    warp = pkg.models.DepthWarpingLayer()
    dcl = pkg.losses.NormalizedDistanceLoss(height, width)
    d1 = pkg.synthetic.smooth_depth(batch_size, height, width, seed=1).to(dev).requires_grad_(True)
    d2 = pkg.synthetic.smooth_depth(batch_size, height, width, seed=2).to(dev).requires_grad_(True)

    def warp_both():
        w21, i1 = warp([d1, d2, batch["boundaries"], batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"], batch["intrinsics"]])
        w12, i2 = warp([d2, d1, batch["boundaries"], batch["translations_2_wrt_1"], batch["rotations_2_wrt_1"], batch["intrinsics"]])
        loss = dcl([d1, w21, i1, batch["intrinsics"]]) + dcl([d2, w12, i2, batch["intrinsics"]])
        d1.grad = d2.grad = None
        loss.backward()

    def warp_both_fused():          # the same chain as ONE library call (endo_warp_consistency): what the metric is quoted on
        with torch.no_grad():
            return pkg.losses.warp_consistency(d1, d2, batch["boundaries"], batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"],
                                               batch["translations_2_wrt_1"], batch["rotations_2_wrt_1"], batch["intrinsics"],
                                               dcl_weight=2.0)

    reps = 20

    def host_inclusive(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        tw = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - tw) / reps / batch_size * 1e3

    warp_ms_per_pair_modules = host_inclusive(warp_both)
    warp_ms_per_pair = host_inclusive(warp_both_fused)
    # device time of the same call: ONE pair of events around `reps` back-to-back calls on the launch stream (round 3 put events
    # around every entry point of the composed chain, which serialised its 11 launches and read SLOWER than the host-inclusive
    # figure); algorithmic bytes as SURVEY.md 8(d) counts them: 160 B per pixel and pair (both directions, forward and backward)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        warp_both_fused()
    ev1.record()
    torch.cuda.synchronize()
    warp_kernel_ms = ev0.elapsed_time(ev1) / reps
    warp_bytes = float(lib.endo_warp_consistency_bytes(batch_size, height, width))          # the library's own count (160 B per pixel of a pair)

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

    if world > 1:
        torch.distributed.barrier()
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    pairs = batch_size * world * args.steps
    if dominant is None:
        dominant = max(MFMA_FAMILIES, key=lambda f: fam[f][0])
    ms, cnt, fl, by = fam[dominant]
    achieved = fl / ms / 1e9 if ms > 0 else 0.0                         # TFLOP/s
    dom_name = lib.endo_prof_family_name(dominant).decode()
    traffic, traffic_src = (None, None)
    if args.config in (1, 2, 4, 5):
        traffic, traffic_src = pmc_traffic(dom_name, args.config)      # bytes per step -> per launch with the launches counted here
    if traffic is not None:
        traffic = traffic / (seen[dominant] / args.steps) if seen[dominant] else None
    warp_gbs = warp_bytes / warp_kernel_ms / 1e6 if warp_kernel_ms > 0 else 0.0
    result = {
        "metric": cfg["metric"],
        "value": pairs / elapsed,
        "unit": "frame-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16 storage and MFMA operands, f32 accumulate" if fp16_storage else "bf16 storage and MFMA operands, f32 accumulate" if bf16_storage else (
            "bf16 MFMA operands, f32 accumulate, f32 storage" if bf16 else "f32"),
        "data": "synthetic",
        "config": {"workload": cfg["workload"], "baseline_config_index": args.config,
                   "global_batch": batch_size * world, "height": height, "width": width, "parallelism": "dp%d" % world},
        "world": world,
        "shared_gpu_plumbing_check_not_a_measurement": True if shared else None,
        "collective_backend": torch.distributed.get_backend() if world > 1 else None,
        "per_rank_pairs_per_s": per_rank,
        "skipped_steps": skipped,
        "kernel_options_overridden": args.kernel_option or None,
        "dispatches_per_step": dispatches_per_step(args.config),
        "conv_roofline_frac_whole_step": (pairs / elapsed) * pair_gflop / 1e3 / (FP32_MFMA_PEAK_TFLOPS * world),
        "depth_warp_fwd_bwd_ms_per_pair": warp_ms_per_pair,
        "roofline": {"kernel": dom_name, "bound": "mfma", "achieved": achieved,
                     "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                     "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC 2*FETCH_SIZE+WRITE_SIZE)",
                     "traffic_source": traffic_src, "algorithmic_bytes_per_launch": by / cnt if cnt else None,
                     "launches": seen[dominant], "timed_launches": cnt, "launches_per_step": lps or None,
                     "sampling": None if period == 1 else "1 launch in %d over the first %d steps: every launch of the step timed %d times" % (
                         period, sampled_steps, sampled_steps // period),
                     "avg_launch_ms": ms / cnt if cnt else None,
                     "family_ms_per_step": (ms / cnt) * lps if cnt and lps else None,
                     # the algorithmic work the timed launches add up to per step, beside SURVEY.md Appendix B's figure for one dense-layer
                     # family (12 244.8 MMAC x 2 x 16 frames at 256 x 320, batch 8; scaled to this config's pixels and batch)
                     "family_gflop_per_step": {"implied_by_timed_launches": (fl / cnt) * lps / 1e9 if cnt and lps else None,
                                               "survey": 391.8336 * (height * width) / (256.0 * 320.0) * batch_size / 8.0},
                     "algorithmic_gbs": by / ms / 1e6 if ms > 0 else None,
                     "concurrent": True,
                     # the dense-layer families one kernel at a time (the warm-up steps with the side stream off): ms per step and TFLOP/s
                     "serial_families_ms_per_step": None if fam_warm is None else {
                         lib.endo_prof_family_name(f).decode(): fam_warm[f][0] / serial_steps for f in MFMA_FAMILIES},
                     "serial_families_tflops": None if fam_warm is None else {
                         lib.endo_prof_family_name(f).decode(): fam_warm[f][2] / fam_warm[f][0] / 1e9 for f in MFMA_FAMILIES}},
        "roofline_serial": None if fam_warm is None else {
            "note": "stand-alone kernel durations: %d warm-up steps with the weight-gradient side stream off (ENDO_OPT_WGRAD_OVERLAP = 0), followed by %d warm-up steps in the product configuration" % (serial_steps, product_steps),
            "kernel": dom_name, "achieved": fam_warm[dominant][2] / fam_warm[dominant][0] / 1e9,
            "frac": fam_warm[dominant][2] / fam_warm[dominant][0] / 1e9 / FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "families_ms_per_step": {lib.endo_prof_family_name(f).decode(): fam_warm[f][0] / serial_steps for f in MFMA_FAMILIES},
            "families_tflops": {lib.endo_prof_family_name(f).decode(): fam_warm[f][2] / fam_warm[f][0] / 1e9 for f in MFMA_FAMILIES}},
        "roofline_depth_warp": {
            "kernel": "depth_warp fwd+bwd (both directions) + depth-consistency loss fwd+bwd", "bound": "hbm",
            "achieved": warp_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": warp_gbs / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_pair": warp_bytes / batch_size,
            "time_base": "device time of endo_warp_consistency (1 memset + 2 kernels), one event pair around %d back-to-back calls" % reps,
            "device_ms_per_pair": warp_kernel_ms / batch_size,
            "host_inclusive_ms_per_pair": warp_ms_per_pair,
            "host_inclusive_ms_per_pair_through_the_modules_and_autograd": warp_ms_per_pair_modules},
    }
    if bf16 or bf16_storage:
        # with bf16 operands the matrix work is ~1/8 of the fp32 kernels' and the dense-layer families are bound by HBM / LDS / VALU:
        # the roofline that applies is HBM (~90 FLOP/B against a bf16 ridge of ~310)
        r = result["roofline"]
        gbs = r["algorithmic_gbs"] or 0.0
        r.update({"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                  "mfma_tflops_for_reference": achieved})
        result["conv_roofline_frac_whole_step"] = None
        if bf16_storage:
            r["algorithmic_bytes_note"] = ("bf16 tensors: dense forward 2 B x (Cin + 12) per pixel; data gradient 2 B x 12 + per input "
                                           "channel the forward value (2 B) and the gradient read and written (4 B); weight gradient "
                                           "2 B x (Cin + 12) per pixel")
            # stand-alone: the same family with the weight gradients in line (warm-up steps), as HBM GB/s of algorithmic bytes
            sgbs = fam_warm[dominant][3] / fam_warm[dominant][0] / 1e6 if fam_warm is not None and fam_warm[dominant][0] > 0 else None
            if result["roofline_serial"] is not None:          # None with --warmup 0 / 1: no stand-alone timing was taken
                result["roofline_serial"].update({"bound": "hbm", "mfma_tflops_for_reference": result["roofline_serial"]["achieved"],
                                                  "achieved": sgbs, "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                                  "frac": sgbs / HBM_PEAK_GBS if sgbs else None})
                result["roofline_serial"].pop("families_tflops", None)
        else:
            result["roofline_serial"] = None
    if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only
        result["cpu_baseline"] = cpu_baseline(cfg)
    print(json.dumps(result))
    sys.stdout.flush()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
