"""Drop-in replacements for the loss modules of reference ``losses.py`` used on the training path
(SparseMaskedL1Loss, NormalizedDistanceLoss -- train.py:210-211) plus ScaleInvariantLoss, on HIP
kernels.  ``forward(x)`` takes ONE list argument, as in the reference (losses.py:22-23, 62-63,
122-123).  Per-sample sums are reduced with wave shuffles + one fp64 atomic per block; the batch
mean happens in a one-wave finalize kernel.
"""

import torch
from torch import nn

from . import _lib


class _SparseL1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flows, flows_hat, masks, eps):
        lib = _lib.load()
        flows = _lib.dev_f32(flows, "flows")
        flows_hat = _lib.dev_f32(flows_hat, "flows from depth")
        masks = _lib.dev_f32(masks, "sparse masks")
        n, c, h, w = flows.shape
        loss = torch.empty((), dtype=torch.float32, device=flows.device)
        stats = torch.empty((n, 2), dtype=torch.float64, device=flows.device)
        _lib.check(lib.endo_sparse_l1_fwd(_lib.ptr(flows), _lib.ptr(flows_hat), _lib.ptr(masks), _lib.ptr(loss), _lib.ptr(stats),
                                          n, c, h * w, eps, _lib.stream()), "endo_sparse_l1_fwd")
        ctx.save_for_backward(flows, flows_hat, masks, stats)
        ctx.eps = eps
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _lib.load()
        flows, flows_hat, masks, stats = ctx.saved_tensors
        n, c, h, w = flows.shape
        grad_loss = _lib.dev_f32(grad_loss, "grad")
        g_f = torch.empty_like(flows) if ctx.needs_input_grad[0] else None
        g_h = torch.empty_like(flows_hat) if ctx.needs_input_grad[1] else None
        _lib.check(lib.endo_sparse_l1_bwd(_lib.ptr(grad_loss), _lib.ptr(flows), _lib.ptr(flows_hat), _lib.ptr(masks),
                                          _lib.ptr(stats), _lib.ptr(g_f), _lib.ptr(g_h), n, c, h * w, ctx.eps, _lib.stream()),
                   "endo_sparse_l1_bwd")
        return g_f, g_h, None, None


class SparseMaskedL1Loss(nn.Module):
    """reference losses.py:57-66."""

    def __init__(self, epsilon=1.0):
        super().__init__()
        self.epsilon = float(epsilon)

    def forward(self, x):
        flows, flows_from_depth, sparse_masks = x
        return _SparseL1Fn.apply(flows, flows_from_depth, sparse_masks, self.epsilon)


class _NormDistFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, warped, intersect, intrinsics, eps):
        lib = _lib.load()
        depth = _lib.dev_f32(depth, "depth maps")
        warped = _lib.dev_f32(warped, "warped depth maps")
        intersect = _lib.dev_f32(intersect, "intersect masks")
        n, _, h, w = depth.shape
        k = _lib.dev_f32(intrinsics, "intrinsics").reshape(n, 9)
        loss = torch.empty((), dtype=torch.float32, device=depth.device)
        stats = torch.empty((n, 4), dtype=torch.float64, device=depth.device)
        _lib.check(lib.endo_norm_dist_fwd(_lib.ptr(depth), _lib.ptr(warped), _lib.ptr(intersect), _lib.ptr(k), _lib.ptr(loss),
                                          _lib.ptr(stats), n, h, w, eps, _lib.stream()), "endo_norm_dist_fwd")
        ctx.save_for_backward(depth, warped, intersect, k, stats)
        ctx.eps = eps
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _lib.load()
        depth, warped, intersect, k, stats = ctx.saved_tensors
        n, _, h, w = depth.shape
        grad_loss = _lib.dev_f32(grad_loss, "grad")
        g_d = torch.empty_like(depth) if ctx.needs_input_grad[0] else None
        g_w = torch.empty_like(warped) if ctx.needs_input_grad[1] else None
        _lib.check(lib.endo_norm_dist_bwd(_lib.ptr(grad_loss), _lib.ptr(depth), _lib.ptr(warped), _lib.ptr(intersect), _lib.ptr(k),
                                          _lib.ptr(stats), _lib.ptr(g_d), _lib.ptr(g_w), n, h, w, ctx.eps, _lib.stream()),
                   "endo_norm_dist_bwd")
        return g_d, g_w, None, None, None


class NormalizedDistanceLoss(nn.Module):
    """reference losses.py:112-146.  ``height`` / ``width`` are accepted for signature parity; the
    pixel grid is generated inside the kernel."""

    def __init__(self, height, width, eps=1.0e-5):
        super().__init__()
        self.height, self.width, self.eps = int(height), int(width), float(eps)

    def forward(self, x):
        depth_maps, warped_depth_maps, intersect_masks, intrinsics = x
        if depth_maps.shape[2] != self.height or depth_maps.shape[3] != self.width:
            raise RuntimeError("NormalizedDistanceLoss was built for %dx%d" % (self.height, self.width))
        return _NormDistFn.apply(depth_maps, warped_depth_maps, intersect_masks, intrinsics, self.eps)


class _ScaleInvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, goal, boundaries, eps):
        lib = _lib.load()
        pred = _lib.dev_f32(pred, "predicted depths")
        goal = _lib.dev_f32(goal, "goal depths")
        boundaries = _lib.dev_f32(boundaries, "boundaries")
        n, hw = pred.shape[0], pred.shape[1] * pred.shape[2] * pred.shape[3]
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        stats = torch.empty((n, 3), dtype=torch.float64, device=pred.device)
        _lib.check(lib.endo_scale_inv_fwd(_lib.ptr(pred), _lib.ptr(goal), _lib.ptr(boundaries), _lib.ptr(loss), _lib.ptr(stats), n,
                                          hw, eps, _lib.stream()), "endo_scale_inv_fwd")
        ctx.save_for_backward(pred, goal, boundaries, stats)
        ctx.eps = eps
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _lib.load()
        pred, goal, boundaries, stats = ctx.saved_tensors
        n, hw = pred.shape[0], pred.shape[1] * pred.shape[2] * pred.shape[3]
        grad_loss = _lib.dev_f32(grad_loss, "grad")
        g_p = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        g_g = torch.empty_like(goal) if ctx.needs_input_grad[1] else None
        _lib.check(lib.endo_scale_inv_bwd(_lib.ptr(grad_loss), _lib.ptr(pred), _lib.ptr(goal), _lib.ptr(boundaries), _lib.ptr(stats),
                                          _lib.ptr(g_p), _lib.ptr(g_g), n, hw, ctx.eps, _lib.stream()), "endo_scale_inv_bwd")
        return g_p, g_g, None, None


class ScaleInvariantLoss(nn.Module):
    """reference losses.py:17-32."""

    def __init__(self, epsilon=1.0e-8):
        super().__init__()
        self.epsilon = float(epsilon)

    def forward(self, x):
        predicted_depths, goal_depths, boundaries = x
        return _ScaleInvFn.apply(predicted_depths, goal_depths, boundaries, self.epsilon)


_consistency_ws = {}


def warp_consistency(depth_maps_1, depth_maps_2, img_masks, translations_1_wrt_2, rotations_1_wrt_2, translations_2_wrt_1,
                     rotations_2_wrt_1, intrinsic_matrices, dcl_weight=1.0, epsilon=1.0e-8):
    """Depth warping both ways + NormalizedDistanceLoss both ways, forward and backward, as one library call
    (``endo_warp_consistency``; reference models.py:454-554 and losses.py:112-146 twice each, train.py:305-314, and the autograd
    backward of that chain):

        loss = dcl_weight * 0.5 * (NDL([d1, warp(d2 -> 1), ...]) + NDL([d2, warp(d1 -> 2), ...]))

    Returns ``(loss, d loss / d depth_maps_1, d loss / d depth_maps_2)``.  The same kernels and arithmetic as
    ``DepthWarpingLayer`` + ``NormalizedDistanceLoss`` under autograd (tests/test_gpu_parity.py::test_warp_consistency_call),
    without the ~10 autograd nodes around them -- the chain BASELINE.json's second metric times."""
    lib = _lib.load()
    d1 = _lib.dev_f32(depth_maps_1, "depth maps 1")
    d2 = _lib.dev_f32(depth_maps_2, "depth maps 2")
    mask = _lib.dev_f32(img_masks, "image masks")
    n, _, h, w = d1.shape
    pose = lambda t, cols, what: _lib.dev_f32(t, what).reshape(n, cols)
    t12, r12 = pose(translations_1_wrt_2, 3, "translations"), pose(rotations_1_wrt_2, 9, "rotations")
    t21, r21 = pose(translations_2_wrt_1, 3, "translations"), pose(rotations_2_wrt_1, 9, "rotations")
    k = pose(intrinsic_matrices, 9, "intrinsics")
    need = int(lib.endo_warp_consistency_workspace_floats(n, h, w))
    key = (d1.device, n, h, w)
    ws = _consistency_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _consistency_ws[key] = torch.empty(need, dtype=torch.float32, device=d1.device)
    loss = torch.empty((), dtype=torch.float32, device=d1.device)
    g1, g2 = torch.empty_like(d1), torch.empty_like(d2)
    _lib.check(lib.endo_warp_consistency(_lib.ptr(d1), _lib.ptr(d2), _lib.ptr(mask), _lib.ptr(t12), _lib.ptr(r12), _lib.ptr(t21),
                                         _lib.ptr(r21), _lib.ptr(k), float(dcl_weight), float(epsilon), _lib.ptr(loss), _lib.ptr(g1),
                                         _lib.ptr(g2), _lib.ptr(ws), n, h, w, _lib.stream()), "endo_warp_consistency")
    return loss, g1, g2
