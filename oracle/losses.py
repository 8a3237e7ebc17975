"""Oracle (test infrastructure): self-supervised losses, CPU torch fp32.

  * sparse_masked_l1     <- reference losses.py:57-66   (SparseMaskedL1Loss)
  * normalized_distance  <- reference losses.py:112-146 (NormalizedDistanceLoss)
  * scale_invariant      <- reference losses.py:17-32   (ScaleInvariantLoss)
Every loss reduces per sample over (C, H, W) and then takes the batch mean.
"""

import torch

from .geometry import pixel_grid

_DIMS = (1, 2, 3)


def sparse_masked_l1(flows, flows_from_depth, sparse_masks, epsilon=1.0):
    """losses.py:62-66: mean_b( sum m |f - f^| / (eps + sum m) ); the numerator spans both
    flow channels, the denominator counts each pixel once."""
    num = (sparse_masks * torch.abs(flows - flows_from_depth)).sum(_DIMS)
    return torch.mean(num / (epsilon + sparse_masks.sum(_DIMS)))


def normalized_distance(depth, warped_depth, intersect, intrinsics, eps=1.0e-5):
    """losses.py:122-146: L1 distance between the two back-projected point maps, normalised by
    the masked depth mass; the 1e-5 * mean term carries no gradient."""
    _, _, h, w = depth.shape
    xs, ys = pixel_grid(h, w, depth.dtype)
    fx = intrinsics[:, 0, 0].reshape(-1, 1, 1, 1)
    fy = intrinsics[:, 1, 1].reshape(-1, 1, 1, 1)
    cx = intrinsics[:, 0, 2].reshape(-1, 1, 1, 1)
    cy = intrinsics[:, 1, 2].reshape(-1, 1, 1, 1)
    with torch.no_grad():
        mean_value = (intersect * depth).sum(_DIMS) / (eps + intersect.sum(_DIMS))
    ax = (xs - cx) / fx
    ay = (ys - cy) / fy
    here = torch.cat([ax * depth, ay * depth, depth], dim=1)
    there = torch.cat([ax * warped_depth, ay * warped_depth, warped_depth], dim=1)
    num = (intersect * torch.abs(here - there)).sum(_DIMS)
    den = 1.0e-5 * mean_value + (intersect * (depth + torch.abs(warped_depth))).sum(_DIMS)
    return torch.mean(2.0 * num / den)


def scale_invariant(predicted, goal, boundaries, epsilon=1.0e-8):
    """losses.py:22-32: r = log(b p + eps) - log(b g + eps);
    mean_b( sum r^2 / sum b + (sum r)^2 / (sum b)^2 )."""
    r = torch.log(boundaries * predicted + epsilon) - torch.log(boundaries * goal + epsilon)
    weight = boundaries.sum(_DIMS)
    first = (r * r).sum(_DIMS) / weight
    total = r.sum(_DIMS)
    return torch.mean(first + total * total / (weight * weight))
