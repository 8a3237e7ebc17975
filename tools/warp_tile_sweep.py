#!/usr/bin/env python3
"""LDS source-tile sweep of the depth-warp kernels (BASELINE.json configs[3]: "LDS tile-size sweep for warp kernel").

For 256x320 (batch 8) and 512x640 (batch 4), every tile shape of endo_depth_warp_{fwd,bwd}_tiled and the L2-gather
kernels (0x0): HIP-event time of forward and backward (both directions of a pair, kernels only, endo_prof family
"geometry") and the achieved GB/s against the algorithmic 7.9 MB per pair at 256x320 (x4 at 512x640).
Also a large-motion case (poses x8), where source boxes stop fitting the staging buffers and blocks fall back to gathers.

    python tools/warp_tile_sweep.py > profiles/r02_warp_tile_sweep.txt
"""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
lib = ea._lib.load()
dev = torch.device("cuda:0")
TILES = [(0, 0), (8, 32), (16, 32), (16, 64), (32, 32), (32, 64)]
FAMILY_GEOMETRY = 10


def prof_ms():
    ms, cnt, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    lib.endo_prof_read(FAMILY_GEOMETRY, ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(fl), ctypes.byref(by))
    return ms.value, cnt.value


def run(n, h, w, pose_scale, reps=30):
    batch = {k: v.to(dev) for k, v in ea.synthetic.make_batch(n, h, w, seed=5, gap_scale=None).items()}
    if pose_scale != 1.0:
        for k in ("translations_1_wrt_2", "translations_2_wrt_1"):
            batch[k] = batch[k] * pose_scale
    d1 = ea.synthetic.smooth_depth(n, h, w, seed=1).to(dev).requires_grad_(True)
    d2 = ea.synthetic.smooth_depth(n, h, w, seed=2).to(dev).requires_grad_(True)
    cot = torch.randn(n, 1, h, w, device=dev)
    rows = []
    ref = None
    for tile in TILES:
        layer = ea.DepthWarpingLayer(tile=tile)

        def both():
            w21, i1 = layer([d1, d2, batch["boundaries"], batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"], batch["intrinsics"]])
            w12, i2 = layer([d2, d1, batch["boundaries"], batch["translations_2_wrt_1"], batch["rotations_2_wrt_1"], batch["intrinsics"]])
            d1.grad = d2.grad = None
            ((w21 * cot).sum() + (w12 * cot).sum()).backward()
            return w21, i1

        for _ in range(3):
            out = both()
        torch.cuda.synchronize()
        lib.endo_prof_enable(1 << FAMILY_GEOMETRY)
        for _ in range(reps):
            both()
        torch.cuda.synchronize()
        ms, cnt = prof_ms()
        lib.endo_prof_enable(0)
        if ref is None:
            ref = (out[0].clone(), out[1].clone(), d1.grad.clone())
        same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
        gerr = float((d1.grad - ref[2]).abs().max() / ref[2].abs().max())
        us_pair = ms / reps / n * 1e3
        mb_pair = 7.9 * (h * w) / (256.0 * 320.0)
        rows.append((tile, us_pair, mb_pair / us_pair * 1e3, same, gerr))
    return rows


def main():
    print("# depth-warp LDS source-tile sweep (tools/warp_tile_sweep.py): forward + backward, both directions of a pair, kernel time")
    print("# (HIP events around the four entry-point calls of a pair; algorithmic bytes 7.9 MB per pair at 256x320, x4 at 512x640)")
    print("# %s" % torch.cuda.get_device_name(0))
    for (n, h, w, scale, what) in ((8, 256, 320, 1.0, "configs[1] size"), (4, 512, 640, 1.0, "configs[3] size"),
                                   (4, 512, 640, 8.0, "configs[3] size, translations x8 (large motion: gather fallback)")):
        print("\n## batch %d, %d x %d -- %s" % (n, h, w, what))
        print("%-10s %12s %10s %22s %18s" % ("tile", "us / pair", "GB/s", "forward == gather", "d1-grad rel diff"))
        for tile, us, gbs, same, gerr in run(n, h, w, scale):
            print("%-10s %12.2f %10.1f %22s %18.2e" % ("%dx%d" % tile if tile != (0, 0) else "gather", us, gbs, same, gerr))


if __name__ == "__main__":
    main()
