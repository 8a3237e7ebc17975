"""One training iteration on the MI355X path -- the counterpart of the batch-loop body of reference
train.py:272-328, built from the drop-in modules of ``models`` / ``losses`` and the fused optimizer.

  colours * boundary -> two network forwards (BN statistics per call, train.py:276-277) -> depth
  scaling -> flow-from-depth both ways -> boundary masking -> sparse-flow loss -> depth warping
  both ways -> depth-consistency loss -> weighted sum (+ the non-finite flag, on the device) -> backward
  -> ONE all-reduce of gradients + flag -> fused clip_grad_norm_(10) + SGD(0.9), skipped by the kernel when
  any rank's loss was NaN / Inf.
"""

import torch

from . import _lib, distributed, losses, models


class _MaskMulFn(torch.autograd.Function):
    """a[n,c,h,w] * mask[n,1,h,w] (train.py:272-273, 293-298); the mask carries no gradient."""

    @staticmethod
    def forward(ctx, a, mask):
        lib = _lib.load()
        a = _lib.dev_f32(a, "tensor")
        mask = _lib.dev_f32(mask, "mask")
        n, c, h, w = a.shape
        out = torch.empty_like(a)
        _lib.check(lib.endo_mask_mul(_lib.ptr(a), _lib.ptr(mask), _lib.ptr(out), n, c, h * w, _lib.stream()), "endo_mask_mul")
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, grad):
        lib = _lib.load()
        (mask,) = ctx.saved_tensors
        grad = _lib.dev_f32(grad, "grad")
        n, c, h, w = grad.shape
        out = torch.empty_like(grad)
        _lib.check(lib.endo_mask_mul(_lib.ptr(grad), _lib.ptr(mask), _lib.ptr(out), n, c, h * w, _lib.stream()), "endo_mask_mul")
        return out, None


def mask_mul(a, mask):
    return _MaskMulFn.apply(a, mask)


class TrainingStep(object):
    def __init__(self, model, optimizer, height, width, sfl_weight=20.0, dcl_weight=0.1, epsilon=1.0e-8, pair_forward=True,
                 fused_head=True, bf16_storage=False, fp16_storage=False):
        self.model = model
        # the network over bf16 level buffers (FCDenseNet.forward_bf16_storage: activations and inter-layer gradients stored as
        # bf16, bf16 matrix cores, fp32 accumulation / statistics / parameter gradients; BASELINE configs[2]); the two frames are two
        # sample groups of one call (each with its own BatchNorm statistics, as the reference's two calls, train.py:276-277); losses,
        # clipping and SGD stay fp32
        # fp16_storage: the same family over IEEE half (FCDenseNet.forward_fp16_storage; BASELINE configs[4]'s storage half)
        self.half_storage = bool(fp16_storage)
        self.bf16_storage = bool(bf16_storage) or self.half_storage
        if self.bf16_storage and not (fused_head and pair_forward):
            raise ValueError("bf16_storage runs through the fused loss head")
        self.epsilon = float(epsilon)
        # everything between the network outputs and d loss / d prediction as one library call (endo_loss_head: the modules'
        # kernels, composed in C) instead of ~60 autograd nodes; needs the grouped pair forward
        self.fused_head = bool(fused_head) and bool(pair_forward) and hasattr(model, "forward_pair_packed")
        self._head_ws = None
        # both frames of the pair through the network as one grouped batch (FCDenseNet.forward_pair): same values as the
        # reference's two calls (train.py:276-277), half the kernel launches
        self.pair_forward = bool(pair_forward) and hasattr(model, "forward_pair")
        self.optimizer = optimizer
        self.sfl_weight = float(sfl_weight)
        self.dcl_weight = float(dcl_weight)
        self.depth_scaling_layer = models.DepthScalingLayer(epsilon=epsilon)
        self.depth_warping_layer = models.DepthWarpingLayer(epsilon=epsilon)
        self.flow_from_depth_layer = models.FlowfromDepthLayer()
        self.sparse_flow_loss_function = losses.SparseMaskedL1Loss()
        self.depth_consistency_loss_function = losses.NormalizedDistanceLoss(height=height, width=width)
        self.bucket = distributed.GradientBucket(model.flat_gradients, getattr(model, "flat_gradient_bucket", None))
        # persistent replicas must start from the same state (nn.DataParallel re-broadcasts on every forward, train.py:197)
        distributed.sync_parameters(model, optimizer)

    def losses(self, batch):
        """Forward part: returns (loss, depth_consistency_loss, sparse_flow_loss, extras)."""
        b = batch["boundaries"]
        colors_1 = mask_mul(batch["colors_1"], b)
        colors_2 = mask_mul(batch["colors_2"], b)
        if self.pair_forward:
            pred_1, pred_2 = self.model.forward_pair(colors_1, colors_2)
        else:
            pred_1 = self.model(colors_1)
            pred_2 = self.model(colors_2)
        scaled_1, std_1 = self.depth_scaling_layer([pred_1, batch["sparse_depths_1"], batch["sparse_depth_masks_1"]])
        scaled_2, std_2 = self.depth_scaling_layer([pred_2, batch["sparse_depths_2"], batch["sparse_depth_masks_2"]])
        flows_1 = self.flow_from_depth_layer([scaled_1, b, batch["translations_1_wrt_2"], batch["rotations_1_wrt_2"],
                                              batch["intrinsics"]])
        flows_2 = self.flow_from_depth_layer([scaled_2, b, batch["translations_2_wrt_1"], batch["rotations_2_wrt_1"],
                                              batch["intrinsics"]])
        with torch.no_grad():
            sparse_flow_masks_1 = mask_mul(batch["sparse_flow_masks_1"], b)
            sparse_flow_masks_2 = mask_mul(batch["sparse_flow_masks_2"], b)
            sparse_flows_1 = mask_mul(batch["sparse_flows_1"], b)
            sparse_flows_2 = mask_mul(batch["sparse_flows_2"], b)
        flows_1 = mask_mul(flows_1, b)
        flows_2 = mask_mul(flows_2, b)
        sfl = self.sfl_weight * 0.5 * (
            self.sparse_flow_loss_function([sparse_flows_1, flows_1, sparse_flow_masks_1]) +
            self.sparse_flow_loss_function([sparse_flows_2, flows_2, sparse_flow_masks_2]))
        warped_21, inter_1 = self.depth_warping_layer([scaled_1, scaled_2, b, batch["translations_1_wrt_2"],
                                                       batch["rotations_1_wrt_2"], batch["intrinsics"]])
        warped_12, inter_2 = self.depth_warping_layer([scaled_2, scaled_1, b, batch["translations_2_wrt_1"],
                                                       batch["rotations_2_wrt_1"], batch["intrinsics"]])
        dcl = self.dcl_weight * 0.5 * (
            self.depth_consistency_loss_function([scaled_1, warped_21, inter_1, batch["intrinsics"]]) +
            self.depth_consistency_loss_function([scaled_2, warped_12, inter_2, batch["intrinsics"]]))
        extras = {"pred_1": pred_1, "pred_2": pred_2, "scaled_1": scaled_1, "scaled_2": scaled_2,
                  "warped_21": warped_21, "warped_12": warped_12, "inter_1": inter_1, "inter_2": inter_2,
                  "std_1": std_1, "std_2": std_2}
        return dcl + sfl, dcl, sfl, extras

    def _fused_iteration(self, batch):
        """Network forward -> endo_loss_head (loss values and d loss / d prediction).  Everything is issued straight through the
        C ABI, without autograd nodes: the caller differentiates the network with ``_fused_backward`` after its guard, which
        saves the autograd engine's start-up latency (~0.1 ms of idle GPU after the loss synchronisation).
        Returns (losses tensor [total, dcl, sfl], network input, forward tape, d loss / d prediction)."""
        lib = _lib.load()
        b = _lib.dev_f32(batch["boundaries"], "boundaries")
        c1 = _lib.dev_f32(batch["colors_1"], "colors_1")
        c2 = _lib.dev_f32(batch["colors_2"], "colors_2")
        n, ch, h, w = c1.shape
        with torch.no_grad():
            x = torch.empty((2 * n, ch, h, w), dtype=torch.float32, device=c1.device)          # both frames, masked (train.py:272-273)
            _lib.check(lib.endo_mask_mul(_lib.ptr(c1), _lib.ptr(b), _lib.ptr(x[:n]), n, ch, h * w, _lib.stream()), "endo_mask_mul")
            _lib.check(lib.endo_mask_mul(_lib.ptr(c2), _lib.ptr(b), _lib.ptr(x[n:]), n, ch, h * w, _lib.stream()), "endo_mask_mul")
            if self.bf16_storage:
                pred, tape = self.model._run_forward16(x, 2, self.half_storage)       # both frames as two sample groups of one call
            else:
                pred, tape = self.model._run_forward(x, 2)          # (2N, 1, H, W): frame 1's predictions first
            need = int(lib.endo_loss_head_workspace_floats(n, h, w))
            if self._head_ws is None or self._head_ws.numel() < need or self._head_ws.device != pred.device:
                self._head_ws = torch.empty(need, dtype=torch.float32, device=pred.device)
            losses_t = torch.empty(4, dtype=torch.float32, device=pred.device)          # total, dcl, sfl, guard flag
            grad_pred = torch.empty_like(pred)
            f = lambda key: _lib.ptr(_lib.dev_f32(batch[key], key))
            pose = lambda key, cols: _lib.ptr(_lib.dev_f32(batch[key], key).reshape(n, cols))
            _lib.check(lib.endo_loss_head(
                _lib.ptr(pred[:n]), _lib.ptr(pred[n:]), _lib.ptr(b), f("sparse_depths_1"), f("sparse_depths_2"),
                f("sparse_depth_masks_1"), f("sparse_depth_masks_2"), f("sparse_flows_1"), f("sparse_flows_2"),
                f("sparse_flow_masks_1"), f("sparse_flow_masks_2"), pose("translations_1_wrt_2", 3), pose("rotations_1_wrt_2", 9),
                pose("translations_2_wrt_1", 3), pose("rotations_2_wrt_1", 9), pose("intrinsics", 9),
                self.sfl_weight, self.dcl_weight, self.epsilon, _lib.ptr(losses_t), _lib.ptr(grad_pred[:n]), _lib.ptr(grad_pred[n:]),
                _lib.ptr(self._head_ws), n, h, w, _lib.stream()), "endo_loss_head")
        return losses_t, x, tape, pred, grad_pred

    def _fused_backward(self, x, tape, grad_pred):
        with torch.no_grad():
            if self.bf16_storage:
                self.model._run_backward16(tuple(x.shape), tape, grad_pred, self.model.training, 2, self.half_storage)
            else:
                self.model._run_backward(x, tape, grad_pred, self.model.training, 2)

    def __call__(self, batch, lr=None):
        """One iteration.  Nothing in it waits for the host: the non-finite-loss guard (train.py:317-322) is a flag the loss head writes on
        the device, summed over ranks inside the gradient all-reduce and read by the optimizer kernel, which then leaves parameters and
        momentum untouched -- the reference's guarded branch also runs backward() and a step() that changes nothing.  Returns a
        ``StepOutput``: a mapping with the keys "loss", "dcl", "sfl", "grad_norm", "skipped" whose first access makes the one
        device-to-host read of the step (the reference's ``loss.item()``, train.py:317) -- a training loop that reads the PREVIOUS
        iteration's output after launching the current one never idles the GPU."""
        if lr is not None:
            for group in self.optimizer.param_groups:
                group["lr"] = lr
        self.optimizer.zero_grad()
        if self.fused_head:
            losses_t, x, tape, pred, grad_pred = self._fused_iteration(batch)
            self._fused_backward(x, tape, grad_pred)
        else:
            loss, dcl, sfl, _ = self.losses(batch)
            with torch.no_grad():
                bad = (~torch.isfinite(loss.detach())).to(torch.float32).reshape(1)
                losses_t = torch.cat([loss.detach().reshape(1), dcl.detach().reshape(1), sfl.detach().reshape(1), bad]).to(torch.float32)
            loss.backward()
        scale, flag = self.bucket.all_reduce(losses_t[3:4])
        norm = self.optimizer.step(grad_scale=scale, skip_flag=flag)
        return StepOutput(losses_t, flag, norm, self._readback_slot(losses_t.device))

    def _readback_slot(self, device):
        """A set of pinned host buffers + an event out of a small ring: the numbers of a step are copied out asynchronously right
        behind the optimizer kernel, so that reading step k - 1 after launching step k waits for step k - 1 only."""
        if device.type != "cuda":
            return None
        ring = self.__dict__.setdefault("_readback_ring", [])
        if len(ring) < 8:
            ring.append(_ReadbackSlot())
            return ring[-1]
        self._readback_next = (self.__dict__.get("_readback_next", -1) + 1) % len(ring)
        slot = ring[self._readback_next]
        slot.release()          # an output issued 8 steps ago and never read takes its values now (its copy finished long ago)
        return slot


class _ReadbackSlot(object):
    def __init__(self):
        self.losses = torch.empty(4, dtype=torch.float32).pin_memory()
        self.flag = torch.empty(1, dtype=torch.float32).pin_memory()
        self.norm = torch.empty(1, dtype=torch.float64).pin_memory()
        self.event = torch.cuda.Event()
        self.owner = None

    def release(self):
        owner = self.owner() if self.owner is not None else None
        if owner is not None:
            owner._read()
        self.owner = None


class StepOutput(object):
    """What one TrainingStep call produced: [total, dcl, sfl] losses, the guard flag after the ranks' consensus and the pre-clip
    gradient norm, copied to pinned host buffers asynchronously right behind the step's last kernel (three copies of a few bytes, no
    kernel).  Read like the dict older versions returned; the first read waits for THOSE copies, not for whatever was queued after
    them.  "loss" is a float; "dcl" / "sfl" / "grad_norm" are 0-dim host tensors (NaN for a skipped step's terms, as before);
    "skipped" is a bool."""

    def __init__(self, losses, flag, norm, slot):
        import weakref
        self._host = None
        self._slot = slot
        if slot is None:          # host tensors (CPU tests of the glue): copies -- with world > 1 `flag` is a view of the gradient bucket's trailing slot, which the next step overwrites before a one-step-late read
            self._vals = (losses.detach().clone(), flag.detach().clone(), norm.detach().clone())
        else:
            slot.losses.copy_(losses, non_blocking=True)
            slot.flag.copy_(flag.reshape(1), non_blocking=True)
            slot.norm.copy_(norm.reshape(1), non_blocking=True)
            slot.event.record()
            slot.owner = weakref.ref(self)

    def _read(self):
        if self._host is None:
            if self._slot is not None:
                self._slot.event.synchronize()
                losses, flag, norm = self._slot.losses.tolist(), self._slot.flag.tolist(), self._slot.norm.tolist()
                self._slot.owner = None
                self._slot = None
            else:
                losses, flag, norm = (t.reshape(-1).tolist() for t in self._vals)
                self._vals = None
            skipped = flag[0] != 0.0
            nan = float("nan")
            self._host = {"loss": losses[0], "dcl": torch.tensor(nan if skipped else losses[1]), "sfl": torch.tensor(nan if skipped else losses[2]),
                          "grad_norm": torch.tensor(norm[0], dtype=torch.float64), "skipped": skipped}
        return self._host

    def __getitem__(self, key):
        return self._read()[key]

    def __contains__(self, key):
        return key in ("loss", "dcl", "sfl", "grad_norm", "skipped")

    def keys(self):
        return self._read().keys()

    def items(self):
        return self._read().items()

    def __repr__(self):
        return "StepOutput(%r)" % (self._read(),)
