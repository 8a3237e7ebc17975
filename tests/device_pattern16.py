"""Test infrastructure: the ACTIVATION PATTERN of a bf16-storage forward pass (endo_net16_fwd), read off its tape -- the counterpart
of device_pattern.py for the 32-channel-blocked bf16 level buffers.  The kernels compute z = fma(x, scale, shift) in fp32 with
scale = gamma * rstd and shift = fma(-mean, scale, beta) (bf16_conv_kernels.h) from the stored bf16 x and the saved fp32 (mean, rstd);
the sign of the exact x * scale + shift is the sign of its fp32 rounding.  Channel order: a level buffer keeps [skip | transition-up
output | up maps]; the reference concatenates [transition-up output, skip] (models.py:183), which is the order of the masks here."""
import importlib

import numpy as np
import torch

ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")

from device_pattern import bn_layers as bn_layers32


def skip(level):
    return 96 + 48 * level


def bn_layers():
    """[(module prefix, level, [buffer channel of reference input channel k])] of the 49 BN layers in module order."""
    out = []
    for prefix, lvl, _, cnt in bn_layers32():
        if prefix.startswith("denseBlocksUp"):
            s = skip(lvl)
            chans = [s + k if k < 48 else (k - 48 if k < s + 48 else k) for k in range(cnt)]
        else:
            chans = list(range(cnt))
        out.append((prefix, lvl, chans))
    return out


def level_buffer(lib, hnd, buf, what, level, n, h, w, half=False):
    """fp32 [n][t][h >> level][w >> level] copy of a level buffer of the tape (what 3) or the gradient workspace (what 4)."""
    if half:
        class _H(object):          # the half build's entry points under the bf16 names used below
            endo_net16_offset = lib.endo_net16h_offset
            endo_bf16_unpack_nhwc = lib.endo_f16_unpack_nhwc
        lib = _H
    t = int(lib.endo_net16_offset(hnd, 5, level))
    hh, ww = h >> level, w >> level
    out = torch.empty((n, t, hh, ww), dtype=torch.float32, device=buf.device)
    rc = lib.endo_bf16_unpack_nhwc(buf.data_ptr() + int(lib.endo_net16_offset(hnd, what, level)), out.data_ptr(), n, t, hh, ww, t, 32, 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    return out.cpu()


def pattern_from_tape(model, tape, n, h, w, half=False):
    lib = ea._lib.load()
    hnd, _, _ = model._handle16(n, h, w, 1, half)
    offset = lib.endo_net16h_offset if half else lib.endo_net16_offset
    params = dict(model.named_parameters())
    raw = tape.detach().cpu().numpy()
    levels = [level_buffer(lib, hnd, tape, 3, lvl, n, h, w, half) for lvl in range(6)]
    pat = {}
    for index, (prefix, lvl, chans) in enumerate(bn_layers()):
        cnt = len(chans)
        off = int(offset(hnd, 1, index))
        saved = torch.from_numpy(raw[off:off + 8 * cnt].view(np.float32).copy()).view(cnt, 2)          # indexed by reference channel
        mean, rstd = saved[:, 0], saved[:, 1]
        gamma = params[prefix + ".weight"].detach().cpu().float()
        beta = params[prefix + ".bias"].detach().cpu().float()
        scale = gamma * rstd                                                                 # fp32 product
        shift = (beta.double() - mean.double() * scale.double()).float()                    # fma(-mean, scale, beta) rounded to fp32
        x = levels[lvl][:, chans]
        z = x.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
        pat["relu::" + prefix] = z > 0
    for lvl in range(5):
        c = skip(lvl)
        hh, ww = h >> (lvl + 1), w >> (lvl + 1)
        off = int(offset(hnd, 2, lvl))
        codes = torch.from_numpy(raw[off:off + n * c * hh * ww].copy()).view(n, hh, ww, c).permute(0, 3, 1, 2).contiguous()
        pat["pool::transDownBlocks.%d" % lvl] = codes
    off = int(offset(hnd, 0, 0))
    pre = torch.from_numpy(raw[off:off + 4 * n * h * w].view(np.float32).copy()).view(n, 1, h, w)
    pat["sign"] = torch.sign(pre)
    return pat


def pattern_of(output, model, n, h, w):
    node = output.grad_fn
    while node is not None and not hasattr(node, "tape"):
        nxt = [fn for fn, _ in node.next_functions if fn is not None]
        node = nxt[0] if nxt else None
    assert node is not None and node.tape is not None, "no forward tape behind this tensor"
    return pattern_from_tape(model, node.tape, n, h, w, bool(getattr(node, "half", False)))
