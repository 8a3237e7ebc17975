"""Test infrastructure: read the ACTIVATION PATTERN of a HIP forward pass off its tape.

A ReLU-BatchNorm network is piecewise linear in its discontinuous choices: which ReLU inputs are positive (reference
models.py:24), which pixel each 2x2 max-pool keeps (models.py:66), the sign under the final |.| (models.py:186).  Two
finite-precision evaluations of the same network disagree on a few of those bits -- wherever a pre-activation lies
within rounding of zero (measured at 2 x 128 x 160: three elements of denseBlocksUp.4's layer-1 maps have |z| < 1e-6,
tests/diag/gpu_diag7.py) -- and every such bit is an O(1) change of that pixel's gradient, for any fp32 implementation,
the reference's CPU path included.  Comparing gradients on the SAME pattern removes that lottery: the oracle's
``forward(..., pattern=...)`` evaluates, in fp64, exactly the branch the HIP pass took, and the HIP gradients must then
agree to fp32 rounding.

The masks are reproduced bit-exactly: the kernels compute z = fma(x - mean, gamma * rstd, beta) in fp32 from the saved
fp32 (mean, rstd) (conv_dma_kernels.h / dgrad_block_kernels.h); the products and sums below are exact in fp64, and
rounding a non-zero exact value to fp32 never changes its sign.
"""
import importlib

import numpy as np
import torch

ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")


def bn_layers():
    """[(module prefix, level, first level-buffer channel, channels)] of the 49 BN layers in module order (net.hip layout)."""
    out = []
    for l in range(5):
        for j in range(4):
            out.append(("denseBlocksDown.%d.layers.%d.norm" % (l, j), l, 48, 48 + 48 * l + 12 * j))
    for l in range(5):
        out.append(("transDownBlocks.%d.norm" % l, l, 48, 96 + 48 * l))
    for j in range(4):
        out.append(("bottleneck.bottleneck.layers.%d.norm" % j, 5, 0, 288 + 12 * j))
    for i in range(5):
        l = 4 - i
        for j in range(4):
            out.append(("denseBlocksUp.%d.layers.%d.norm" % (i, j), l, 0, 144 + 48 * l + 12 * j))
    return out


def pattern_from_tape(model, tape, n, h, w, groups=1):
    """One pattern dict per sample group of a forward pass over ``groups * n`` samples (n per group)."""
    lib = ea._lib.load()
    hnd, _, _ = model._handle(n, h, w, groups)
    stride = int(lib.endo_net_group_stride(hnd)) if groups > 1 else 0
    params = dict(model.named_parameters())
    raw = tape.detach().cpu()
    patterns = []
    for g in range(groups):
        t = raw[g * stride:] if groups > 1 else raw
        pat = {}
        levels = []
        for lvl in range(6):
            ch = lib.endo_net_level_channels(lvl)
            off = lib.endo_net_act_offset(hnd, lvl)
            hh, ww = h >> lvl, w >> lvl
            levels.append(t[off:off + n * ch * hh * ww].view(n, ch, hh, ww))
        for index, (prefix, lvl, c0, cnt) in enumerate(bn_layers()):
            off = lib.endo_net_tape_offset(hnd, 1, index)
            saved = t[off:off + 2 * cnt].view(cnt, 2)
            mean, rstd = saved[:, 0].contiguous(), saved[:, 1].contiguous()
            gamma = params[prefix + ".weight"].detach().cpu().float()
            beta = params[prefix + ".bias"].detach().cpu().float()
            scale = gamma * rstd                                        # fp32 product, as the kernels form it
            xcen = levels[lvl][:, c0:c0 + cnt] - mean.view(1, -1, 1, 1)  # fp32 subtraction, as the kernels do
            z = xcen.double() * scale.double().view(1, -1, 1, 1) + beta.double().view(1, -1, 1, 1)
            pat["relu::" + prefix] = z > 0
        tape_bytes = t.numpy().view(np.uint8)
        for lvl in range(5):
            c = 96 + 48 * lvl
            hh, ww = h >> (lvl + 1), w >> (lvl + 1)
            off = lib.endo_net_tape_offset(hnd, 2, lvl)
            codes = torch.from_numpy(tape_bytes[off:off + n * c * hh * ww].copy()).view(n, c, hh, ww)
            pat["pool::transDownBlocks.%d" % lvl] = codes
        off = lib.endo_net_tape_offset(hnd, 0, 0)
        pre = t[off:off + n * h * w].view(n, 1, h, w)
        pat["sign"] = torch.sign(pre)
        patterns.append(pat)
    return patterns


def pattern_of(output, model, n, h, w, groups=1):
    """Pattern(s) of the forward pass that produced ``output`` (a tensor returned by model(x) / forward_pair with autograd on)."""
    node = output.grad_fn
    while node is not None and not hasattr(node, "tape"):
        nxt = [fn for fn, _ in node.next_functions if fn is not None]
        node = nxt[0] if nxt else None
    assert node is not None and node.tape is not None, "no forward tape behind this tensor"
    return pattern_from_tape(model, node.tape, n, h, w, groups)
