"""Diagnostic (not a test): after FCDenseNet57.forward_bf16_storage + backward, compare every convolution output's TOTAL gradient in
the bf16 gradient workspace (endo_net16_bwd) with the autograd gradient of the same map in the bf16-rounding oracle
(oracle.network.forward(quant=bf16_ste, pattern=the pass's own, trace=...)), in backward order: the first map that disagrees locates a defect.
usage: python tests/diag/bf16_bwd_trace.py [train|eval] [n h w]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
from oracle import network as onet

mode = sys.argv[1] if len(sys.argv) > 1 else "train"
n, h, w = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (1, 64, 96)
dev = torch.device("cuda:0")
state = onet.keep_depth_positive(onet.perturb_affine(onet.synthetic_state(71), 72))
rng = np.random.default_rng(29)
x = torch.from_numpy(rng.uniform(-1, 1, (n, 3, h, w)).astype(np.float32))
g = torch.from_numpy(rng.standard_normal((n, 1, h, w)).astype(np.float32))

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from device_pattern16 import pattern_of

m = ea.FCDenseNet57(1)
m.load_state_dict(state)
m = m.to(dev)
getattr(m, mode)()
y = m.forward_bf16_storage(x.to(dev))
pattern = pattern_of(y, m, n, h, w)
y.backward(g.to(dev))
torch.cuda.synchronize()

st64 = {k: (v.double().requires_grad_(k in onet.trainable_names()) if v.is_floating_point() else v.clone()) for k, v in state.items()}
trace = {}
y64 = onet.forward(st64, x.double(), training=(mode == "train"), quant=onet.bf16_ste, trace=trace, pattern=pattern)
for v in trace.values():
    if v.requires_grad:
        v.retain_grad()
y64.backward(g.double())
ws = m._gradws[("bf16", n, h, w, 1)]
lib = ea._lib.load()

T = [192, 256, 288, 352, 384, 352]
down_in = lambda l: 48 + 48 * l
skip = lambda l: down_in(l) + 48
offs, o = [], 0
for l in range(6):
    offs.append(o)
    o += (n * (h >> l) * (w >> l) * T[l] * 2 + 255) // 256 * 256


def dbuf(level, c0, count):
    hh, ww = h >> level, w >> level
    out = torch.empty((n, count, hh, ww), dtype=torch.float32, device=dev)
    rc = lib.endo_bf16_unpack_nhwc(ws.data_ptr() + offs[level], out.data_ptr(), n, count, hh, ww, T[level], 32, c0, None)
    assert rc == 0
    torch.cuda.synchronize()
    return out.double().cpu()


maps = []
for i in range(5):
    l = 4 - i                                     # denseBlocksUp.i lives at level 4 - i ... in forward order i = 0 is the coarsest
for i in reversed(range(5)):
    l = 4 - i
    for j in reversed(range(4)):
        maps.append(("conv::denseBlocksUp.%d.layers.%d" % (i, j), l, skip(l) + 48 + 12 * j, 12))
    maps.append(("conv::transUpBlocks.%d" % i, l, skip(l), 48))
for j in reversed(range(4)):
    maps.append(("conv::bottleneck.bottleneck.layers.%d" % j, 5, 288 + 12 * j, 12))
for l in reversed(range(5)):
    maps.append(("conv::transDownBlocks.%d" % l, l + 1, 0, skip(l)))
    for j in reversed(range(4)):
        maps.append(("conv::denseBlocksDown.%d.layers.%d" % (l, j), l, down_in(l) + 12 * j, 12))
maps.append(("conv::firstconv", 0, 0, 48))
print("forward: max |y16 - y oracle| / max = %.2e" % float((y.detach().double().cpu() - y64.detach()).abs().max() / y64.detach().abs().max()))
for name, level, c0, count in maps:
    want = trace[name].grad
    got = dbuf(level, c0, count)
    err = float((got - want).abs().max() / want.abs().max())
    l2 = float((got - want).norm() / want.norm())
    # coherent parts of the error, per channel: its mean over the pixels (an offset) against the channel's rms, and how the sum over
    # the pixels compares (a training-mode BatchNorm makes the true sum of most maps exactly zero)
    d = got - want
    rms = want.pow(2).mean(dim=(0, 2, 3)).sqrt()
    off = float((d.mean(dim=(0, 2, 3)).abs() / rms).max())
    print("%-44s level %d ch %3d +%3d   max err / max %.2e   relative L2 %.2e   worst channel offset / rms %.2e   (max |g| %.2e)" % (
        name, level, c0, count, err, l2, off, float(want.abs().max())))
