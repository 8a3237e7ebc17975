"""Oracle (test infrastructure): differentiable geometry layers, CPU torch fp32.

Restates, from the math in SURVEY.md Appendix A:
  * depth_scaling        <- reference models.py:339-363  (DepthScalingLayer.forward)
  * flow_from_depth      <- reference models.py:377-451  (_warp_coordinate_generate, _flow_from_depth)
  * depth_warping        <- reference models.py:469-554  (_depth_warping) + models.py:325-336
                            (_bilinear_interpolate == F.grid_sample(bilinear, zeros,
                            align_corners=False) on the grid (2u/W-1, 2v/H-1))
All tensors are NCHW fp32.  Gradients come from torch autograd on these ops, so
the oracle is also the reference for every backward kernel.
"""

import torch
import torch.nn.functional as F


def pixel_grid(height, width, dtype=torch.float32):
    """(x, y) pixel-coordinate planes, shape (H, W) each -- models.py:381-386 ('ij' meshgrid)."""
    ys = torch.arange(height, dtype=dtype).reshape(height, 1).expand(height, width)
    xs = torch.arange(width, dtype=dtype).reshape(1, width).expand(height, width)
    return xs, ys


def camera_maps(intrinsics, rotations, translations):
    """Per-sample M = K R^T K^-1 (N,3,3) and w = -K R^T t (N,3) -- models.py:391-399 / 492-499.

    The reference obtains K^-1 with an LU solve against the identity; so do we.
    """
    n = intrinsics.shape[0]
    eye = torch.eye(3, dtype=intrinsics.dtype).expand(n, 3, 3)
    k_inv = torch.linalg.solve(intrinsics, eye)
    k_rt = torch.bmm(intrinsics, rotations.transpose(1, 2))
    w = torch.bmm(k_rt, -translations.reshape(n, 3, 1)).reshape(n, 3)
    m = torch.bmm(k_rt, k_inv)
    return m, w, k_inv


def _rays(m, height, width):
    """q = M (x, y, 1)^T for every pixel -> (N, 3, H, W) -- models.py:401-402."""
    xs, ys = pixel_grid(height, width, m.dtype)
    p = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1).reshape(1, height, width, 3, 1)
    q = torch.matmul(m.reshape(-1, 1, 1, 3, 3), p).reshape(-1, height, width, 3)
    return q.permute(0, 3, 1, 2)


def depth_scaling(pred, sparse_depth, sparse_mask, epsilon=1.0e-8):
    """models.py:346-363.  Returns (scale * pred, mean_b(std/mean of the sparse scale map))."""
    dims = (1, 2, 3)
    binary = (sparse_mask > 1.0e-8).to(pred.dtype)
    mean_sd = (sparse_depth * binary).sum(dims, keepdim=True) / binary.sum(dims, keepdim=True)
    above = (sparse_depth > 0.5 * mean_sd).to(pred.dtype)
    smap = sparse_depth * above / (epsilon + pred)
    count = above.sum(dims, keepdim=True)
    mean_scale = smap.sum(dims, keepdim=True) / count
    centered = smap - above * mean_scale
    std = torch.sqrt((centered * centered).sum(dims) / above.sum(dims))
    scale = smap.sum(dims) / above.sum(dims)
    return scale.reshape(-1, 1, 1, 1) * pred, torch.mean(std / mean_scale)


def projected_coordinates(depth, mask, translations, rotations, intrinsics):
    """(u2, v2) of frame-1 pixels in frame 2, the flow-layer variant (no z>0 clamp).

    models.py:404-429: z2 = w_z + d q_z ; z2~ = 1e30 (1-m) + m z2 ; u2 = (w_x + d q_x)/z2~.
    """
    n, _, h, w_ = depth.shape
    m, w, _ = camera_maps(intrinsics, rotations, translations)
    q = _rays(m, h, w_)
    wv = w.reshape(n, 3, 1, 1)
    z2 = wv[:, 2:3] + depth * q[:, 2:3]
    z2 = 1.0e30 * (1.0 - mask) + mask * z2
    u2 = (wv[:, 0:1] + depth * q[:, 0:1]) / z2
    v2 = (wv[:, 1:2] + depth * q[:, 1:2]) / z2
    return u2, v2


def flow_from_depth(depth, mask, translations, rotations, intrinsics):
    """models.py:433-451: flow = ((u2 - x)/W, (v2 - y)/H) as N x 2 x H x W."""
    _, _, h, w_ = depth.shape
    u2, v2 = projected_coordinates(depth, mask, translations, rotations, intrinsics)
    xs, ys = pixel_grid(h, w_, depth.dtype)
    return torch.cat([(u2 - xs) / float(w_), (v2 - ys) / float(h)], dim=1)


def bilinear_sample(image, u, v):
    """models.py:325-336: grid = (2u/W - 1, 2v/H - 1), grid_sample defaults of torch 2.10
    (bilinear, zeros, align_corners=False) => samples at (u - 0.5, v - 0.5)."""
    _, _, h, w_ = image.shape
    grid = torch.stack([2.0 * (u[:, 0] / float(w_)) - 1.0, 2.0 * (v[:, 0] / float(h)) - 1.0], dim=-1)
    return F.grid_sample(image, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def depth_warping_parts(depth_1, depth_2, mask, translations, rotations, intrinsics, epsilon=1.0e-8):
    """models.py:473-552 up to (but not including) the 0.9 threshold.

    Returns (warped, sampled_mask * mask); tests use the second value to exclude
    pixels that sit numerically on the threshold.
    """
    n, _, h, w_ = depth_1.shape
    d1 = depth_1 * mask
    d2 = depth_2 * mask
    m, w, k_inv = camera_maps(intrinsics, rotations, translations)
    q = _rays(m, h, w_)
    wv = w.reshape(n, 3, 1, 1)
    z2 = wv[:, 2:3] + d1 * q[:, 2:3]
    eps = torch.tensor(epsilon, dtype=d1.dtype)
    z2 = torch.where(mask > 0.5, z2, eps)
    z2 = torch.where(z2 > 0.0, z2, eps)
    u2 = (wv[:, 0:1] + d1 * q[:, 0:1]) / z2
    v2 = (wv[:, 1:2] + d1 * q[:, 1:2]) / z2

    # depth 2 re-expressed as camera-1 depth on frame 2's own grid (models.py:531-541)
    w2 = torch.bmm(intrinsics, translations.reshape(n, 3, 1)).reshape(n, 3, 1, 1)
    m2 = torch.bmm(torch.bmm(intrinsics, rotations), k_inv)
    s = _rays(m2, h, w_)[:, 2:3]
    d_in_1 = mask * (w2[:, 2:3] + d2 * s)

    warped = bilinear_sample(d_in_1, u2, v2)
    overlap = bilinear_sample(mask, u2, v2) * mask
    return warped, overlap


def depth_warping(depth_1, depth_2, mask, translations, rotations, intrinsics, epsilon=1.0e-8):
    """models.py:460-465 / 469-554: returns [warped depth 2->1, binary intersect mask]."""
    warped, overlap = depth_warping_parts(depth_1, depth_2, mask, translations, rotations,
                                          intrinsics, epsilon)
    intersect = (overlap >= 0.9).to(warped.dtype).detach()
    return warped, intersect
