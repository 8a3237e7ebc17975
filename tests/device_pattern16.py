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


def level_buffer(lib, hnd, buf, what, level, n, h, w, half=False, to_cpu=True):
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
    return out.cpu() if to_cpu else out


def pattern_from_tape(model, tape, n, h, w, half=False, groups=1):
    """n: samples per group.  One pattern per sample group (a list when groups > 1): every group has its own BatchNorm statistics."""
    lib = ea._lib.load()
    hnd, _, _ = model._handle16(n, h, w, groups, half)
    offset = lib.endo_net16h_offset if half else lib.endo_net16_offset
    params = dict(model.named_parameters())
    raw = tape.detach().cpu().numpy()
    nt = n * groups
    group_bytes = int(offset(hnd, 7, 0)) if groups > 1 else 0
    pats = [{} for _ in range(groups)]
    layers = bn_layers()
    # (the sign evaluation below runs where the tape lives -- fp64 torch arithmetic on the device is test plumbing, exact either way)
    for lvl in range(6):          # one level buffer at a time: level 0 of the benchmark batch is 1 GB as fp32
        x_all = level_buffer(lib, hnd, tape, 3, lvl, nt, h, w, half, to_cpu=False)
        for index, (prefix, l, chans) in enumerate(layers):
            if l != lvl:
                continue
            cnt = len(chans)
            gamma = params[prefix + ".weight"].detach().cpu().float()
            beta = params[prefix + ".bias"].detach().cpu().float()
            for g in range(groups):
                off = int(offset(hnd, 1, index)) + g * group_bytes
                saved = torch.from_numpy(raw[off:off + 8 * cnt].view(np.float32).copy()).view(cnt, 2)          # indexed by reference channel
                mean, rstd = saved[:, 0], saved[:, 1]
                scale = gamma * rstd                                                                 # fp32 product
                shift = (beta.double() - mean.double() * scale.double()).float()                    # fma(-mean, scale, beta) rounded to fp32
                x = x_all[g * n:(g + 1) * n][:, torch.tensor(chans, device=x_all.device)]
                z = x.double() * scale.double().view(1, -1, 1, 1).to(x.device) + shift.double().view(1, -1, 1, 1).to(x.device)
                pats[g]["relu::" + prefix] = (z > 0).cpu()
                del x, z
        del x_all
    for lvl in range(5):
        c = skip(lvl)
        hh, ww = h >> (lvl + 1), w >> (lvl + 1)
        off = int(offset(hnd, 2, lvl))
        codes = torch.from_numpy(raw[off:off + nt * c * hh * ww].copy()).view(nt, hh, ww, c).permute(0, 3, 1, 2).contiguous()
        for g in range(groups):
            pats[g]["pool::transDownBlocks.%d" % lvl] = codes[g * n:(g + 1) * n]
    off = int(offset(hnd, 0, 0))
    pre = torch.from_numpy(raw[off:off + 4 * nt * h * w].view(np.float32).copy()).view(nt, 1, h, w)
    for g in range(groups):
        pats[g]["sign"] = torch.sign(pre[g * n:(g + 1) * n])
    return pats if groups > 1 else pats[0]


def pattern_of(output, model, n, h, w):
    node = output.grad_fn
    while node is not None and not hasattr(node, "tape"):
        nxt = [fn for fn, _ in node.next_functions if fn is not None]
        node = nxt[0] if nxt else None
    assert node is not None and node.tape is not None, "no forward tape behind this tensor"
    return pattern_from_tape(model, node.tape, n, h, w, bool(getattr(node, "half", False)))
