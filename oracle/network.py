"""Oracle (test infrastructure): FC-DenseNet57 as a functional CPU torch program.

Restates reference models.py:19-194 (DenseLayer, DenseBlock, TransitionDown, TransitionUp,
Bottleneck, FCDenseNet.forward, FCDenseNet57) over a flat ``{name: tensor}`` dict that uses the
reference's state-dict key names, so a reference checkpoint drops straight in.

  growth 12, 4 layers per block, first conv 48, 5 down + bottleneck + 5 up  (models.py:190-194)
  dense layer  = BN -> ReLU -> conv3x3(bias)                                 (models.py:19-28)
  down block   returns input ++ 4 new maps; up block returns only the new    (models.py:39-53)
  transition down = BN -> ReLU -> conv1x1 -> maxpool2                        (models.py:56-67)
  transition up   = nearest x2 -> conv3x3 -> centre crop -> cat(out, skip)   (models.py:70-80)
  output = |conv1x1(last block)|                                             (models.py:186)
"""

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

GROWTH = 12
LAYERS_PER_BLOCK = 4
FIRST = 48
LEVELS = 5
BN_EPS = 1.0e-5
BN_MOMENTUM = 0.1


def parameter_spec():
    """Ordered [(name, shape, kind)] in the reference's ``.parameters()`` / state-dict order.

    kind: 'conv_w', 'conv_b', 'bn_w', 'bn_b', 'bn_rm', 'bn_rv', 'bn_nbt'.
    """
    spec = []

    def conv(prefix, cout, cin, k):
        spec.append((prefix + ".weight", (cout, cin, k, k), "conv_w"))
        spec.append((prefix + ".bias", (cout,), "conv_b"))

    def bn(prefix, c):
        spec.append((prefix + ".weight", (c,), "bn_w"))
        spec.append((prefix + ".bias", (c,), "bn_b"))
        spec.append((prefix + ".running_mean", (c,), "bn_rm"))
        spec.append((prefix + ".running_var", (c,), "bn_rv"))
        spec.append((prefix + ".num_batches_tracked", (), "bn_nbt"))

    def dense_block(prefix, cin):
        for j in range(LAYERS_PER_BLOCK):
            bn("%s.layers.%d.norm" % (prefix, j), cin + j * GROWTH)
            conv("%s.layers.%d.conv" % (prefix, j), GROWTH, cin + j * GROWTH, 3)

    conv("firstconv", FIRST, 3, 3)
    c = FIRST
    skips = []
    down, trans = [], []
    for i in range(LEVELS):
        down.append(("denseBlocksDown.%d" % i, c))
        c += GROWTH * LAYERS_PER_BLOCK
        skips.append(c)
        trans.append(("transDownBlocks.%d" % i, c))
    for prefix, cin in down:
        dense_block(prefix, cin)
    for prefix, cc in trans:
        bn(prefix + ".norm", cc)
        conv(prefix + ".conv", cc, cc, 1)
    dense_block("bottleneck.bottleneck", c)
    new = GROWTH * LAYERS_PER_BLOCK
    for i in range(LEVELS):
        conv("transUpBlocks.%d.convTrans.1" % i, new, new, 3)
    last = None
    for i in range(LEVELS):
        cin = new + skips[LEVELS - 1 - i]
        dense_block("denseBlocksUp.%d" % i, cin)
        last = cin + new
    conv("finalConv", 1, last, 1)
    return spec


def synthetic_state(seed, dtype=torch.float32):
    """Deterministic Kaiming-normal(fan_in, relu) conv weights, zero biases, BN gamma=1 beta=0,
    running stats (0, 1) -- the distribution of reference utils.py:655-671 (called train.py:193) --
    drawn from a numpy PCG64 stream so fixtures do not depend on torch's RNG."""
    rng = np.random.default_rng(seed)
    state = OrderedDict()
    for name, shape, kind in parameter_spec():
        if kind == "conv_w":
            fan_in = shape[1] * shape[2] * shape[3]
            std = math.sqrt(2.0 / fan_in)
            state[name] = torch.from_numpy((rng.standard_normal(shape) * std).astype(np.float32)).to(dtype)
        elif kind in ("conv_b", "bn_b", "bn_rm"):
            state[name] = torch.zeros(shape, dtype=dtype)
        elif kind in ("bn_w", "bn_rv"):
            state[name] = torch.ones(shape, dtype=dtype)
        else:
            state[name] = torch.zeros(shape, dtype=torch.long)
    return state


def perturb_affine(state, seed):
    """Make BN gamma/beta and conv biases non-trivial so that parity tests exercise them."""
    rng = np.random.default_rng(seed)
    for name, shape, kind in parameter_spec():
        if kind == "bn_w":
            state[name] = torch.from_numpy((1.0 + 0.2 * rng.standard_normal(shape)).astype(np.float32))
        elif kind in ("bn_b", "conv_b"):
            state[name] = torch.from_numpy((0.1 * rng.standard_normal(shape)).astype(np.float32))
    return state


def keep_depth_positive(state, bias=4.0):
    """Shift the final conv's bias so |conv + bias| stays away from zero.  With random weights the
    raw output crosses zero, and DepthScalingLayer divides by it (models.py:356): a 1e-6 difference
    in the prediction then moves the recovered scale by 1e-3.  A trained network predicts positive
    depth everywhere; this keeps the end-to-end parity tests in that regime."""
    state["finalConv.bias"] = state["finalConv.bias"] + bias
    return state


def trainable_names():
    return [n for n, _, k in parameter_spec() if k in ("conv_w", "conv_b", "bn_w", "bn_b")]


def bf16_ste(t):
    """Round to bfloat16 (nearest even) in the forward direction, identity in the backward direction: the ``quant`` argument of
    ``forward`` that restates where the bf16-storage kernel family (endo_net16_fwd) rounds -- the input image, every stored
    convolution output, every staged relu(bn(x)) and every matrix-core weight -- so that autograd yields the exact gradient of THAT
    function, against which endo_net16_bwd differs only by its bf16 storage of the gradients between layers."""
    return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())


def fp16_ste(t):
    """``bf16_ste`` for IEEE half storage (endo_net16h_fwd)."""
    return t + (t.detach().to(torch.float16).to(t.dtype) - t.detach())


def _q(quant, t):
    return t if quant is None else quant(t)


def _bn_relu(state, prefix, x, training, pattern=None, quant=None):
    y = F.batch_norm(x, state[prefix + ".running_mean"], state[prefix + ".running_var"],
                     state[prefix + ".weight"], state[prefix + ".bias"],
                     training, BN_MOMENTUM, BN_EPS)
    if training:
        state[prefix + ".num_batches_tracked"] += 1
    if pattern is not None:
        return _q(quant, y * pattern["relu::" + prefix].to(y.dtype))
    return _q(quant, F.relu(y))


def _max_pool(x, prefix, pattern=None):
    if pattern is None:
        return F.max_pool2d(x, 2)
    n, c, h, w = x.shape
    windows = x.reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
    return torch.gather(windows, 4, pattern["pool::" + prefix].long().unsqueeze(-1)).squeeze(-1)


def _dense_block(state, prefix, x, training, keep_input, trace=None, pattern=None, quant=None):
    new = []
    for j in range(LAYERS_PER_BLOCK):
        p = "%s.layers.%d" % (prefix, j)
        a = _bn_relu(state, p + ".norm", x, training, pattern, quant)
        out = _q(quant, F.conv2d(a, _q(quant, state[p + ".conv.weight"]), state[p + ".conv.bias"], padding=1))
        if trace is not None:
            trace["conv::" + p] = out
        x = torch.cat([x, out], dim=1)
        new.append(out)
    return x if keep_input else torch.cat(new, dim=1)


def forward(state, x, training=True, trace=None, pattern=None, quant=None):
    """FCDenseNet.forward (models.py:171-187).  ``state`` running buffers are updated in place in
    training mode, exactly as nn.BatchNorm2d does.

    ``pattern`` (optional dict) FIXES the network's discontinuous choices instead of deriving them from the values:
    "relu::<bn module>" (bool, the inputs ReLU lets through), "pool::<transition down module>" (which of the 2x2 pixels
    each max-pool keeps, 2 * row + col) and "sign" (the sign under the final |.|).  With the pattern another evaluation
    took -- tests read it off the HIP forward pass -- the result is that evaluation's piecewise-linear branch of the
    network, computed exactly: gradients can then be compared without the O(1) per-pixel differences a single flipped
    ReLU bit makes between ANY two finite-precision runs.

    ``trace`` (optional dict) receives the
    intermediate maps: skip_L (down block L output), bott_in, bott_new, tu_L, upnew_L, and "conv::<module>" = the
    output of every convolution (after the max-pool for a transition down), whose autograd gradient is the TOTAL
    gradient of those maps -- what the HIP gradient workspace holds for the same channel planes."""
    # quant (optional, e.g. bf16_ste): applied to the input, to every stored convolution output, to every relu(bn(x)) fed to a
    # convolution and to the weights of every convolution but the final 1 x 1 (which the bf16-storage family evaluates in fp32)
    out = _q(quant, F.conv2d(_q(quant, x), _q(quant, state["firstconv.weight"]), state["firstconv.bias"], padding=1))
    if trace is not None:
        trace["conv::firstconv"] = out
    skips = []
    for i in range(LEVELS):
        out = _dense_block(state, "denseBlocksDown.%d" % i, out, training, keep_input=True, trace=trace, pattern=pattern, quant=quant)
        skips.append(out)
        if trace is not None:
            trace["skip_%d" % i] = out
        p = "transDownBlocks.%d" % i
        a = _bn_relu(state, p + ".norm", out, training, pattern, quant)
        out = _q(quant, _max_pool(F.conv2d(a, _q(quant, state[p + ".conv.weight"]), state[p + ".conv.bias"]), p, pattern))
        if trace is not None:
            trace["conv::" + p] = out
    if trace is not None:
        trace["bott_in"] = out
    out = _dense_block(state, "bottleneck.bottleneck", out, training, keep_input=False, trace=trace, pattern=pattern, quant=quant)
    if trace is not None:
        trace["bott_new"] = out
    for i in range(LEVELS):
        skip = skips.pop()
        p = "transUpBlocks.%d.convTrans.1" % i
        up = F.interpolate(out, scale_factor=2, mode="nearest")
        up = _q(quant, F.conv2d(up, _q(quant, state[p + ".weight"]), state[p + ".bias"], padding=1))
        dy = (up.shape[2] - skip.shape[2]) // 2
        dx = (up.shape[3] - skip.shape[3]) // 2
        up = up[:, :, dy:dy + skip.shape[2], dx:dx + skip.shape[3]]
        if trace is not None:
            trace["tu_%d" % (LEVELS - 1 - i)] = up
            trace["conv::transUpBlocks.%d" % i] = up
        out = torch.cat([up, skip], dim=1)
        out = _dense_block(state, "denseBlocksUp.%d" % i, out, training, keep_input=(i == LEVELS - 1), trace=trace,
                           pattern=pattern, quant=quant)
        if trace is not None:
            trace["upnew_%d" % (LEVELS - 1 - i)] = out[:, -GROWTH * LAYERS_PER_BLOCK:]
    pre = F.conv2d(out, state["finalConv.weight"], state["finalConv.bias"])
    if pattern is not None:
        return pre * pattern["sign"].to(pre.dtype)
    return torch.abs(pre)


def conv_macs(height, width):
    """Forward multiply-accumulates per frame (16 098 086 400 at 256x320)."""
    total = 0
    h, w = height, width
    res = {}

    def at(level):
        return (height >> level) * (width >> level)

    total += at(0) * 3 * FIRST * 9
    c = FIRST
    skips = []
    for i in range(LEVELS):
        for j in range(LAYERS_PER_BLOCK):
            total += at(i) * (c + j * GROWTH) * GROWTH * 9
        c += GROWTH * LAYERS_PER_BLOCK
        skips.append(c)
        total += at(i) * c * c
    for j in range(LAYERS_PER_BLOCK):
        total += at(LEVELS) * (c + j * GROWTH) * GROWTH * 9
    new = GROWTH * LAYERS_PER_BLOCK
    for i in range(LEVELS):
        level = LEVELS - 1 - i
        total += at(level) * new * new * 9
        cin = new + skips[level]
        for j in range(LAYERS_PER_BLOCK):
            total += at(level) * (cin + j * GROWTH) * GROWTH * 9
        last = cin + new
    total += at(0) * last
    del h, w, res
    return total
