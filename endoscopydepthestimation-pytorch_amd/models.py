"""Drop-in replacements for the ``nn.Module`` classes of reference ``models.py`` that ``train.py`` /
``evaluate.py`` instantiate, running on hand-written HIP kernels for MI355X (libendo_hip.so).

API kept from the reference:
  * ``FCDenseNet57(n_classes=1)``: ``forward(x: N x 3 x H x W) -> N x 1 x H x W`` (>= 0); same
    ``state_dict`` key names and ``.parameters()`` order as reference models.py:100-194, so reference
    checkpoints load and ``torch.optim.SGD`` / ``clip_grad_norm_`` iterate the same 210 tensors.
  * ``DepthScalingLayer(epsilon)``, ``FlowfromDepthLayer()``, ``DepthWarpingLayer(epsilon)``:
    ``forward(x)`` takes ONE list argument and returns a tensor or a pair (models.py:346-347,
    370-371, 460-461).
Tensors are contiguous NCHW fp32 on the current HIP device; there is no CPU path.

Design notes (DESIGN.md has the full story): parameters are views into one flat fp32 buffer and
their ``.grad`` are views into one flat gradient buffer -- the kernels accumulate weight gradients
there directly, the fused clip+SGD step and the single RCCL all-reduce run over the flat buffers.
"""

import ctypes

import torch
from torch import nn

from . import _lib

GROWTH = 12
LAYERS = 4
FIRST = 48
LEVELS = 5


# ---------------------------------------------------------------------------------------------
# parameter holders (they only own tensors; all arithmetic happens inside the HIP library)
# ---------------------------------------------------------------------------------------------
class ConvParams(nn.Module):
    """weight [cout, cin, k, k] + bias [cout] of one convolution (reference nn.Conv2d slots)."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)


class FusedBatchNorm2d(nn.Module):
    """gamma/beta + running statistics of one BatchNorm2d (eps 1e-5, momentum 0.1).  The
    normalisation itself is fused into the load path of the convolution that follows it."""

    def __init__(self, channels):
        super().__init__()
        self.num_features = channels
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class DenseLayer(nn.Module):          # reference models.py:19-28
    def __init__(self, cin):
        super().__init__()
        self.norm = FusedBatchNorm2d(cin)
        self.conv = ConvParams(cin, GROWTH, 3)


class DenseBlock(nn.Module):          # reference models.py:31-53
    def __init__(self, cin):
        super().__init__()
        self.layers = nn.ModuleList([DenseLayer(cin + j * GROWTH) for j in range(LAYERS)])


class TransitionDown(nn.Module):      # reference models.py:56-67
    def __init__(self, channels):
        super().__init__()
        self.norm = FusedBatchNorm2d(channels)
        self.conv = ConvParams(channels, channels, 1)


class _Nearest2x(nn.Module):
    """Placeholder for nn.Upsample at index 0 of ``convTrans`` (keeps the key 'convTrans.1.*')."""


class TransitionUp(nn.Module):        # reference models.py:70-80
    def __init__(self, channels):
        super().__init__()
        self.convTrans = nn.ModuleList([_Nearest2x(), ConvParams(channels, channels, 3)])


class Bottleneck(nn.Module):          # reference models.py:83-90
    def __init__(self, cin):
        super().__init__()
        self.bottleneck = DenseBlock(cin)


class _NetFunction(torch.autograd.Function):
    """Whole-network forward/backward as ONE autograd node (the reference builds ~400 per call)."""

    @staticmethod
    def forward(ctx, x, anchor, net, groups=1):
        x = _lib.dev_f32(x, "FCDenseNet57 input")
        out, tape = net._run_forward(x, groups)
        ctx.net = net
        ctx.training = net.training
        ctx.tape = tape
        ctx.groups = groups
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("FCDenseNet57 on the MI355X path does not produce a gradient for its input image "
                               "(the training path never needs one, reference train.py:272-277); detach the input")
        if ctx.tape is None:
            raise RuntimeError("FCDenseNet57: the forward tape of this call was released by its first backward pass; "
                               "run the forward again (retain_graph=True is not supported for the network node)")
        (x,) = ctx.saved_tensors
        ctx.net._run_backward(x, ctx.tape, grad_out, ctx.training, ctx.groups)
        ctx.tape = None
        return None, None, None, None


class _Net16Function(torch.autograd.Function):
    """The bf16-storage network (endo_net16_fwd / endo_net16_bwd) as one autograd node."""

    @staticmethod
    def forward(ctx, x, anchor, net, half=False):
        x = _lib.dev_f32(x, "FCDenseNet57 input")
        out, tape = net._run_forward16(x, 1, half)
        ctx.half = half
        ctx.net = net
        ctx.training = net.training
        ctx.tape = tape
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.needs_input_grad[0]:
            raise RuntimeError("FCDenseNet57 on the MI355X path does not produce a gradient for its input image; detach the input")
        if ctx.tape is None:
            raise RuntimeError("FCDenseNet57: the forward tape of this call was released by its first backward pass")
        ctx.net._run_backward16(ctx.shape, ctx.tape, grad_out, ctx.training, 1, ctx.half)
        ctx.tape = None
        return None, None, None, None


class FCDenseNet(nn.Module):
    """FC-DenseNet57 (the only configuration the drivers use, reference models.py:190-194)."""

    def __init__(self, n_classes=1):
        super().__init__()
        if n_classes != 1:
            raise ValueError("the MI355X path implements the depth network (n_classes=1) only")
        self.firstconv = ConvParams(3, FIRST, 3)
        c = FIRST
        skips = []
        self.denseBlocksDown = nn.ModuleList()
        self.transDownBlocks = nn.ModuleList()
        for _ in range(LEVELS):
            self.denseBlocksDown.append(DenseBlock(c))
            c += GROWTH * LAYERS
            skips.insert(0, c)
            self.transDownBlocks.append(TransitionDown(c))
        self.bottleneck = Bottleneck(c)
        new = GROWTH * LAYERS
        self.transUpBlocks = nn.ModuleList()
        self.denseBlocksUp = nn.ModuleList()
        for i in range(LEVELS):
            self.transUpBlocks.append(TransitionUp(new))
            self.denseBlocksUp.append(DenseBlock(new + skips[i]))
        self.finalConv = ConvParams(new + skips[-1] + new, 1, 1)

        self._flat = None          # flat parameter storage (all 210 tensors)
        self._flat_grad = None
        self._flat_grad_full = None
        self._flat_bn = None       # running_mean/var of the 49 BN layers
        self._nbt = None
        self._handles = {}
        self._gradws = {}
        self._anchor = None
        self._flatten()

    # ---- flat storage -------------------------------------------------------------------------
    def _bn_modules(self):
        return [m for m in self.modules() if isinstance(m, FusedBatchNorm2d)]

    def _flatten(self):
        """(Re)pack parameters and BN buffers into flat buffers and make the tensors views of them."""
        params = list(self.parameters())
        device = params[0].device
        total = sum(p.numel() for p in params)
        flat = torch.empty(total, dtype=torch.float32, device=device)
        off = 0
        offsets = []
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1).float())
            p.data = flat[off:off + n].view(p.shape)
            p.grad = None
            offsets.append(off)
            off += n
        self._flat, self._offsets, self._params = flat, offsets, params
        self._flat_grad = None
        self._flat_grad_full = None
        bns = self._bn_modules()
        width = sum(2 * m.num_features for m in bns)
        flat_bn = torch.empty(width, dtype=torch.float32, device=device)
        nbt = torch.empty(len(bns), dtype=torch.long, device=device)
        off = 0
        for i, m in enumerate(bns):
            c = m.num_features
            flat_bn[off:off + c].copy_(m.running_mean)
            flat_bn[off + c:off + 2 * c].copy_(m.running_var)
            nbt[i] = m.num_batches_tracked
            m._buffers["running_mean"] = flat_bn[off:off + c]
            m._buffers["running_var"] = flat_bn[off + c:off + 2 * c]
            m._buffers["num_batches_tracked"] = nbt[i]
            off += 2 * c
        self._flat_bn, self._nbt, self._bns = flat_bn, nbt, bns
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self._gradws = {}

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._flatten()          # .cuda() / .to() re-allocate every tensor: re-pack into flat buffers
        return self

    def _views_intact(self):
        base = self._flat.data_ptr()
        for p, off in zip(self._params, self._offsets):
            if p.data_ptr() != base + 4 * off:
                return False
        off = 0
        base = self._flat_bn.data_ptr()
        for m in self._bns:
            if m.running_mean.data_ptr() != base + 4 * off:
                return False
            off += 2 * m.num_features
        return True

    def flat_parameters(self):
        """The flat fp32 buffer all 210 parameters are views of (reference .parameters() order)."""
        return self._flat

    def flat_gradients(self, create=True):
        """The flat gradient buffer; every ``p.grad`` is a view of it (zero-filled on (re)attach)."""
        self._attach_grads(create)
        return self._flat_grad

    def flat_gradient_bucket(self):
        """The buffer a data-parallel step all-reduces: the flat gradients plus one trailing float, the step's non-finite-loss flag."""
        self._attach_grads(True)
        return self._flat_grad_full

    def _attach_grads(self, create=True):
        if self._flat_grad is None or self._flat_grad.device != self._flat.device:
            if not create:
                return
            # one extra float behind the gradients: the non-finite-loss flag of the step, so that the ONE all-reduce of the bucket also
            # carries the guard's consensus (train_step.TrainingStep, distributed.GradientBucket)
            self._flat_grad_full = torch.zeros(self._flat.numel() + 1, dtype=torch.float32, device=self._flat.device)
            self._flat_grad = self._flat_grad_full[:self._flat.numel()]
            for p, off in zip(self._params, self._offsets):
                p.grad = self._flat_grad[off:off + p.numel()].view(p.shape)
            return
        base = self._flat_grad.data_ptr()
        intact = True
        for p, off in zip(self._params, self._offsets):
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * off:
                intact = False
                break
        if not intact:       # e.g. optimizer.zero_grad(set_to_none=True): start a fresh accumulation
            self._flat_grad.zero_()
            for p, off in zip(self._params, self._offsets):
                p.grad = self._flat_grad[off:off + p.numel()].view(p.shape)

    # ---- HIP library plumbing ------------------------------------------------------------------
    def _handle(self, n, h, w, groups=1):
        """n = samples per group."""
        key = (n, h, w, groups)
        if key not in self._handles:
            lib = _lib.load()
            if lib.endo_net_param_floats() != self._flat.numel():
                raise RuntimeError("parameter packing mismatch between Python and libendo_hip.so")
            hnd = ctypes.c_void_p()
            _lib.check(lib.endo_net_create_grouped(ctypes.byref(hnd), n, h, w, groups), "endo_net_create_grouped(%d,%d,%d,%d)" % key)
            for option_id, value in getattr(self, "_kernel_options", {}).items():
                lib.endo_net_set_option(hnd, option_id, value)
            self._handles[key] = (hnd, int(lib.endo_net_tape_floats(hnd)), int(lib.endo_net_gradws_floats(hnd)))
        return self._handles[key]

    # kernel-form / precision options (include/endo_hip.h ENDO_OPT_*) belong to THIS module object -- like everything else about a
    # reference module (train.py:191): they apply to every native handle it owns, present and future, and to no other model
    _OPTION_DEFAULTS = {0: 5, 1: 3, 2: 2, 3: 1024, 4: 0, 5: 1, 6: 0, 7: 1, 8: 1, 9: 3}          # default_options() of csrc/net.hip

    def set_kernel_option(self, option_id, value):
        """Returns the previous value."""
        option_id, value = int(option_id), int(value)
        if option_id not in self._OPTION_DEFAULTS:
            raise ValueError("unknown kernel option %d" % option_id)
        options = self.__dict__.setdefault("_kernel_options", {})
        old = options.get(option_id, self._OPTION_DEFAULTS[option_id])
        options[option_id] = value
        lib = _lib.load()
        for key, (hnd, _, _) in self._handles.items():
            if key[0] in ("bf16", "fp16"):
                continue
            rc = lib.endo_net_set_option(hnd, option_id, value)
            if rc < 0:
                raise RuntimeError("endo_net_set_option(%d, %d) failed: %d" % (option_id, value, rc))
        return old

    def kernel_option(self, option_id):
        return self.__dict__.get("_kernel_options", {}).get(int(option_id), self._OPTION_DEFAULTS[int(option_id)])

    def __del__(self):
        try:
            lib = _lib.load()
            for key, (hnd, _, _) in self._handles.items():
                {"bf16": lib.endo_net16_destroy, "fp16": lib.endo_net16h_destroy}.get(key[0], lib.endo_net_destroy)(hnd)
            self._handles = {}
        except Exception:
            pass

    # native handles, workspaces and the gradient views belong to ONE module object: a copy (copy.deepcopy for an EMA or
    # best-model snapshot, pickling) starts without them and re-packs its own flat buffers
    def __getstate__(self):
        state = dict(self.__dict__)
        for key in ("_handles", "_gradws"):
            state[key] = {}
        for key in ("_flat_grad", "_flat_grad_full", "_anchor"):
            state[key] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._handles, self._gradws = {}, {}
        self._flatten()

    def __deepcopy__(self, memo):
        import copy
        clone = self.__class__.__new__(self.__class__)
        memo[id(self)] = clone
        state = self.__getstate__()
        for key in ("_flat", "_flat_bn", "_nbt", "_params", "_bns", "_offsets"):
            state.pop(key, None)          # rebuilt by _flatten from the copied parameter / buffer tensors
        clone.__dict__.update(copy.deepcopy(state, memo))
        clone._handles, clone._gradws = {}, {}
        for p in clone.parameters():
            p.grad = None
        clone._flatten()
        return clone

    def _run_forward(self, x, groups=1):
        lib = _lib.load()
        x = _lib.dev_f32(x, "FCDenseNet57 input")
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError("expected N x 3 x H x W input")
        if x.device != self._flat.device:
            raise RuntimeError("model and input are on different devices; call model.cuda() first")
        if not self._views_intact():
            self._flatten()
        n, _, h, w = x.shape
        if n % groups:
            raise RuntimeError("batch of %d samples does not split into %d groups" % (n, groups))
        hnd, tape_floats, _ = self._handle(n // groups, h, w, groups)
        tape = torch.empty(tape_floats, dtype=torch.float32, device=x.device)
        out = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
        _lib.check(lib.endo_net_fwd(hnd, _lib.ptr(self._flat), _lib.ptr(self._flat_bn), _lib.ptr(x), _lib.ptr(out),
                                    _lib.ptr(tape), 1 if self.training else 0, _lib.stream()), "endo_net_fwd")
        if self.training:
            self._nbt.add_(groups)
        return out, tape

    def _run_backward(self, x, tape, grad_out, training, groups=1):
        lib = _lib.load()
        n, _, h, w = x.shape
        hnd, _, gradws_floats = self._handle(n // groups, h, w, groups)
        key = (n, h, w, groups)
        if key not in self._gradws:
            self._gradws[key] = torch.empty(gradws_floats, dtype=torch.float32, device=x.device)
        self._attach_grads()
        grad_out = _lib.dev_f32(grad_out, "grad_output")
        _lib.check(lib.endo_net_bwd(hnd, _lib.ptr(self._flat), _lib.ptr(x), _lib.ptr(tape), _lib.ptr(grad_out),
                                    _lib.ptr(self._flat_grad), _lib.ptr(self._gradws[key]),
                                    1 if training else 0, _lib.stream()), "endo_net_bwd")

    def forward(self, x):
        if torch.is_grad_enabled():
            return _NetFunction.apply(x, self._anchor, self)
        out, _ = self._run_forward(x)
        return out

    def forward_bf16_storage(self, x):
        """The network over bf16 level buffers (``endo_net16_fwd`` / ``endo_net16_bwd``: activations and inter-layer gradients stored
        as bf16 in 32-channel blocks, bf16 matrix cores with fp32 accumulation; BatchNorm statistics, parameter gradients and the
        output in fp32 -- the bf16-storage family of BASELINE configs[2]).  Same parameters, same running-statistics
        semantics (``.train()``: batch statistics and an update; ``.eval()``: running statistics) -- a different function numerically
        (every stored activation is rounded to 8 significant bits), with its own tolerances (tests/test_gpu_bf16.py).  Differentiable
        with respect to the parameters; H and W must be multiples of 32."""
        if torch.is_grad_enabled():
            return _Net16Function.apply(x, self._anchor, self)
        out, _ = self._run_forward16(_lib.dev_f32(x, "FCDenseNet57 input"))
        return out

    def forward_fp16_storage(self, x):
        """``forward_bf16_storage`` over IEEE half level buffers (``endo_net16h_*``: BASELINE configs[4]'s "fp16 storage / fp32
        accumulate"): the same kernels compiled for another element type; the backward pass scales the stored gradients by a power of
        two chosen from max |grad_output| (half's range) and returns unscaled parameter gradients.  Its own bounds: tests/test_gpu_bf16.py."""
        if torch.is_grad_enabled():
            return _Net16Function.apply(x, self._anchor, self, True)
        out, _ = self._run_forward16(_lib.dev_f32(x, "FCDenseNet57 input"), 1, True)
        return out

    @staticmethod
    def _api16(half):
        """the entry points of the 16-bit-storage family for bf16 (endo_net16_*) or half (endo_net16h_*)"""
        lib = _lib.load()
        prefix = "endo_net16h_" if half else "endo_net16_"
        return lambda name: getattr(lib, prefix + name)

    def set_wgrad_overlap16(self, on):
        """The 16-bit-storage family's weight gradients on the backward pass's own side stream (default) or in line on the caller's
        stream (``endo_net16_set_wgrad_overlap``): applies to this module's handles, present and future."""
        # 0: in line; 1 / True: forked after every layer's prep_dy; 2: one fork per dense block
        self.__dict__["_wgrad_overlap16"] = int(on)
        for key, (hnd, _, _) in self._handles.items():
            if key[0] in ("bf16", "fp16"):
                self._api16(key[0] == "fp16")("set_wgrad_overlap")(hnd, int(on))

    def _handle16(self, n, h, w, groups=1, half=False):
        """n: samples per group"""
        api = self._api16(half)
        key = ("fp16" if half else "bf16", n, h, w, groups)
        if key not in self._handles:
            hnd = ctypes.c_void_p()
            _lib.check(api("create")(ctypes.byref(hnd), n, h, w, groups), "endo_net16_create(%d,%d,%d,%d)" % (n, h, w, groups))
            api("set_wgrad_overlap")(hnd, int(self.__dict__.get("_wgrad_overlap16", 1)))
            self._handles[key] = (hnd, int(api("tape_bytes")(hnd)), int(api("bwd_workspace_bytes")(hnd)))
        return self._handles[key]

    def forward_pair_bf16_storage(self, x1, x2):
        """(forward_bf16_storage(x1), forward_bf16_storage(x2)) as ONE pass over a grouped batch (endo_net16 with two sample groups: every
        launch covers both frames, each frame keeps its own BatchNorm batch statistics, the running statistics are updated with x1's
        first) -- what a training step runs.  No autograd node: TrainingStep differentiates it through ``_run_backward16``."""
        x = torch.cat([_lib.dev_f32(x1, "FCDenseNet57 input"), _lib.dev_f32(x2, "FCDenseNet57 input")], dim=0)
        out, _ = self._run_forward16(x, 2)
        n = x1.shape[0]
        return out[:n], out[n:]

    def _run_forward16(self, x, groups=1, half=False):
        api = self._api16(half)
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError("expected N x 3 x H x W input")
        if x.device != self._flat.device:
            raise RuntimeError("model and input are on different devices; call model.cuda() first")
        if not self._views_intact():
            self._flatten()
        n, _, h, w = x.shape
        hnd, tape_bytes, _ = self._handle16(n // groups, h, w, groups, half)
        tape = torch.empty(tape_bytes, dtype=torch.uint8, device=x.device)
        out = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
        with torch.no_grad():
            _lib.check(api("fwd")(hnd, _lib.ptr(self._flat), _lib.ptr(self._flat_bn), _lib.ptr(x), _lib.ptr(out), _lib.ptr(tape),
                                  1 if self.training else 0, _lib.stream()), "endo_net16_fwd")
        if self.training:
            self._nbt.add_(groups)
        return out, tape

    def _run_backward16(self, shape, tape, grad_out, training, groups=1, half=False):
        api = self._api16(half)
        n, _, h, w = shape
        hnd, _, ws_bytes = self._handle16(n // groups, h, w, groups, half)
        key = ("fp16" if half else "bf16", n // groups, h, w, groups)
        if key not in self._gradws:
            self._gradws[key] = torch.empty(ws_bytes, dtype=torch.uint8, device=tape.device)
        self._attach_grads()
        grad_out = _lib.dev_f32(grad_out, "grad_output")
        _lib.check(api("bwd")(hnd, _lib.ptr(self._flat), _lib.ptr(tape), _lib.ptr(grad_out), _lib.ptr(self._flat_grad),
                              _lib.ptr(self._gradws[key]), 1 if training else 0, _lib.stream()), "endo_net16_bwd")

    def forward_pair(self, x1, x2):
        """``(self(x1), self(x2))`` -- the two forward passes of a training step (reference train.py:276-277) -- as ONE
        pass over a grouped batch: every kernel launch covers both frames, each frame keeps its own BatchNorm batch
        statistics and the running statistics are updated first with x1's, then with x2's, so values and gradients
        equal the two separate calls (up to fp32 summation order in the shared parameter gradients).  Half the
        launches, twice the blocks per launch: it is what the coarse levels of the network need on 256 CUs."""
        if x1.shape != x2.shape:
            raise RuntimeError("forward_pair needs two batches of the same shape")
        x = torch.cat([_lib.dev_f32(x1, "FCDenseNet57 input"), _lib.dev_f32(x2, "FCDenseNet57 input")], dim=0)
        if torch.is_grad_enabled():
            out = _NetFunction.apply(x, self._anchor, self, 2)
        else:
            out, _ = self._run_forward(x, 2)
        n = x1.shape[0]
        return out[:n], out[n:]

    def forward_pair_packed(self, x1, x2):
        """``forward_pair`` returning ONE tensor of 2N samples (frame 1's predictions first): the caller that differentiates
        both halves at once (``train_step.TrainingStep``'s fused loss head) hands back one gradient tensor instead of having
        autograd assemble it from two slices."""
        if x1.shape != x2.shape:
            raise RuntimeError("forward_pair needs two batches of the same shape")
        x = torch.cat([_lib.dev_f32(x1, "FCDenseNet57 input"), _lib.dev_f32(x2, "FCDenseNet57 input")], dim=0)
        if torch.is_grad_enabled():
            return _NetFunction.apply(x, self._anchor, self, 2)
        return self._run_forward(x, 2)[0]

    def level_buffers(self, x):
        """Debug/test hook: run a forward and return the six level buffers (views of the tape)."""
        lib = _lib.load()
        out, tape = self._run_forward(x)
        n, _, h, w = x.shape
        hnd, _, _ = self._handle(n, h, w)
        levels = []
        for lvl in range(LEVELS + 1):
            ch = lib.endo_net_level_channels(lvl)
            off = lib.endo_net_act_offset(hnd, lvl)
            hh, ww = h >> lvl, w >> lvl
            levels.append(tape[off:off + n * ch * hh * ww].view(n, ch, hh, ww))
        return out, levels


def FCDenseNet57(n_classes):
    return FCDenseNet(n_classes=n_classes)


# ---------------------------------------------------------------------------------------------
# geometry layers
# ---------------------------------------------------------------------------------------------
class _DepthScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, sparse_depth, sparse_mask, eps):
        lib = _lib.load()
        pred = _lib.dev_f32(pred, "depth estimations")
        sd = _lib.dev_f32(sparse_depth, "sparse depths")
        sm = _lib.dev_f32(sparse_mask, "sparse masks")
        n, hw = pred.shape[0], pred.shape[1] * pred.shape[2] * pred.shape[3]
        scaled = torch.empty_like(pred)
        ratio = torch.empty((), dtype=torch.float32, device=pred.device)
        stats = torch.empty((n, 8), dtype=torch.float64, device=pred.device)
        _lib.check(lib.endo_depth_scale_fwd(_lib.ptr(pred), _lib.ptr(sd), _lib.ptr(sm), _lib.ptr(scaled), _lib.ptr(ratio),
                                            _lib.ptr(stats), n, hw, eps, _lib.stream()), "endo_depth_scale_fwd")
        ctx.save_for_backward(pred, sd, stats)
        ctx.eps = eps
        ctx.set_materialize_grads(False)
        return scaled, ratio

    @staticmethod
    def backward(ctx, grad_scaled, grad_ratio):
        lib = _lib.load()
        pred, sd, stats = ctx.saved_tensors
        if grad_scaled is None and grad_ratio is None:
            return None, None, None, None
        n, hw = pred.shape[0], pred.shape[1] * pred.shape[2] * pred.shape[3]
        gs = None if grad_scaled is None else _lib.dev_f32(grad_scaled, "grad")
        gr = None if grad_ratio is None else _lib.dev_f32(grad_ratio, "grad")
        grad_pred = torch.empty_like(pred)
        work = torch.empty((n,), dtype=torch.float64, device=pred.device)
        _lib.check(lib.endo_depth_scale_bwd(_lib.ptr(gs), _lib.ptr(gr), _lib.ptr(pred), _lib.ptr(sd), _lib.ptr(stats),
                                            _lib.ptr(grad_pred), _lib.ptr(work), n, hw, ctx.eps, _lib.stream()),
                   "endo_depth_scale_bwd")
        return grad_pred, None, None, None


class DepthScalingLayer(nn.Module):
    """reference models.py:339-363: per-sample scale recovery from sparse SfM depths."""

    def __init__(self, epsilon=1.0e-8):
        super().__init__()
        self.epsilon = float(epsilon)

    def forward(self, x):
        absolute_depth_estimations, input_sparse_depths, input_weighted_sparse_masks = x
        return _DepthScaleFn.apply(absolute_depth_estimations, input_sparse_depths, input_weighted_sparse_masks,
                                   self.epsilon)


def _pose(t, r, k, n):
    t = _lib.dev_f32(t, "translation vectors").reshape(n, 3)
    r = _lib.dev_f32(r, "rotation matrices").reshape(n, 9)
    k = _lib.dev_f32(k, "intrinsic matrices").reshape(n, 9)
    return t, r, k


class _FlowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, mask, t, r, k):
        lib = _lib.load()
        depth = _lib.dev_f32(depth, "depth maps")
        mask = _lib.dev_f32(mask, "image masks")
        n, _, h, w = depth.shape
        t, r, k = _pose(t, r, k, n)
        flow = torch.empty((n, 2, h, w), dtype=torch.float32, device=depth.device)
        _lib.check(lib.endo_flow_from_depth_fwd(_lib.ptr(depth), _lib.ptr(mask), _lib.ptr(t), _lib.ptr(r), _lib.ptr(k),
                                                _lib.ptr(flow), n, h, w, _lib.stream()), "endo_flow_from_depth_fwd")
        ctx.save_for_backward(depth, mask, t, r, k)
        return flow

    @staticmethod
    def backward(ctx, grad_flow):
        lib = _lib.load()
        depth, mask, t, r, k = ctx.saved_tensors
        n, _, h, w = depth.shape
        grad_flow = _lib.dev_f32(grad_flow, "grad")
        grad_depth = torch.empty_like(depth)
        _lib.check(lib.endo_flow_from_depth_bwd(_lib.ptr(grad_flow), _lib.ptr(depth), _lib.ptr(mask), _lib.ptr(t), _lib.ptr(r),
                                                _lib.ptr(k), _lib.ptr(grad_depth), n, h, w, _lib.stream()),
                   "endo_flow_from_depth_bwd")
        return grad_depth, None, None, None, None


class FlowfromDepthLayer(nn.Module):
    """reference models.py:366-374: dense flow induced by depth + relative pose."""

    def forward(self, x):
        depth_maps_1, img_masks, translation_vectors, rotation_matrices, intrinsic_matrices = x
        return _FlowFn.apply(depth_maps_1, img_masks, translation_vectors, rotation_matrices, intrinsic_matrices)


class _WarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d1, d2, mask, t, r, k, eps, tile=None):
        lib = _lib.load()
        d1 = _lib.dev_f32(d1, "depth maps 1")
        d2 = _lib.dev_f32(d2, "depth maps 2")
        mask = _lib.dev_f32(mask, "image masks")
        n, _, h, w = d1.shape
        t, r, k = _pose(t, r, k, n)
        warped = torch.empty_like(d1)
        intersect = torch.empty_like(d1)
        if tile is None:
            _lib.check(lib.endo_depth_warp_fwd(_lib.ptr(d1), _lib.ptr(d2), _lib.ptr(mask), _lib.ptr(t), _lib.ptr(r), _lib.ptr(k),
                                               _lib.ptr(warped), _lib.ptr(intersect), n, h, w, eps, _lib.stream()),
                       "endo_depth_warp_fwd")
        else:
            _lib.check(lib.endo_depth_warp_fwd_tiled(_lib.ptr(d1), _lib.ptr(d2), _lib.ptr(mask), _lib.ptr(t), _lib.ptr(r), _lib.ptr(k),
                                                     _lib.ptr(warped), _lib.ptr(intersect), n, h, w, eps, int(tile[0]), int(tile[1]),
                                                     _lib.stream()), "endo_depth_warp_fwd_tiled(%d x %d)" % (tile[0], tile[1]))
        ctx.save_for_backward(d1, d2, mask, t, r, k)
        ctx.eps = eps
        ctx.tile = tile
        ctx.mark_non_differentiable(intersect)
        return warped, intersect

    @staticmethod
    def backward(ctx, grad_warped, _grad_intersect):
        lib = _lib.load()
        d1, d2, mask, t, r, k = ctx.saved_tensors
        n, _, h, w = d1.shape
        grad_warped = _lib.dev_f32(grad_warped, "grad")
        g1 = torch.empty_like(d1)
        g2 = torch.empty_like(d2)
        if ctx.tile is None:
            _lib.check(lib.endo_depth_warp_bwd(_lib.ptr(grad_warped), _lib.ptr(d1), _lib.ptr(d2), _lib.ptr(mask), _lib.ptr(t),
                                               _lib.ptr(r), _lib.ptr(k), _lib.ptr(g1), _lib.ptr(g2), n, h, w, ctx.eps,
                                               _lib.stream()), "endo_depth_warp_bwd")
        else:
            _lib.check(lib.endo_depth_warp_bwd_tiled(_lib.ptr(grad_warped), _lib.ptr(d1), _lib.ptr(d2), _lib.ptr(mask), _lib.ptr(t),
                                                     _lib.ptr(r), _lib.ptr(k), _lib.ptr(g1), _lib.ptr(g2), n, h, w, ctx.eps,
                                                     int(ctx.tile[0]), int(ctx.tile[1]), _lib.stream()), "endo_depth_warp_bwd_tiled")
        return g1, g2, None, None, None, None, None, None


class DepthWarpingLayer(nn.Module):
    """reference models.py:454-465: warp depth map 2 into frame 1 + binary intersection mask.

    ``tile`` (not in the reference; default None = the library's choice): the LDS source-tile shape of the kernels,
    ``(tile_h, tile_w)`` in {(8,32), (16,32), (16,64), (32,32), (32,64)}, or ``(0, 0)`` for the L2-gather kernels.  It
    changes speed only (tools/warp_tile_sweep.py)."""

    def __init__(self, epsilon=1.0e-8, tile=None):
        super().__init__()
        self.epsilon = float(epsilon)
        self.tile = None if tile is None else (int(tile[0]), int(tile[1]))

    def forward(self, x):
        depth_maps_1, depth_maps_2, img_masks, translation_vectors, rotation_matrices, intrinsic_matrices = x
        warped, intersect = _WarpFn.apply(depth_maps_1, depth_maps_2, img_masks, translation_vectors, rotation_matrices,
                                          intrinsic_matrices, self.epsilon, self.tile)
        return warped, intersect
