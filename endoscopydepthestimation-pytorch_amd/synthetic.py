"""Synthetic training batches with the shapes, ranges and sparsity of the reference's data loader.

The reference's ``SfMDataset.__getitem__`` (reference dataset.py:336-462) yields 16 tensors per frame
pair (train.py:244-270).  There is no dataset on the benchmark box, so this module draws batches of
the same shapes from a seeded numpy stream, following SURVEY.md section 8(d):

  colours        U(-1, 1), N x 3 x H x W          (albu.Normalize(mean .5, std .5), dataset.py:148)
  boundary       binary octagon, ~59 % ones        (endoscope mask, dataset.py:427-430)
  intrinsics     fx = fy = 169.29275, cx = 130.03175, cy = 106.9795 at 256 x 320, scaled with size
  poses          axis-angle ~ N(0, 0.02^2), t ~ N(0, 0.03^2); inverse pose as dataset.py:398-399
  sparse depth   500 pixels per plane inside the mask, U(0.2, 0.8); mask = 1 there
  sparse flow    N(0, 0.03^2) at the same pixels; flow mask = same support

Everything is produced as contiguous fp32 CPU tensors; callers move them to the device.
"""

import numpy as np
import torch

BATCH_KEYS = ("colors_1", "colors_2", "sparse_depths_1", "sparse_depths_2",
              "sparse_depth_masks_1", "sparse_depth_masks_2", "sparse_flows_1", "sparse_flows_2",
              "sparse_flow_masks_1", "sparse_flow_masks_2", "boundaries",
              "rotations_1_wrt_2", "rotations_2_wrt_1", "translations_1_wrt_2",
              "translations_2_wrt_1", "intrinsics")


def boundary_mask(height, width):
    """Centred octagon covering ~59 % of the frame (the example mask has 59.4 % ones)."""
    ys = (np.arange(height, dtype=np.float64) + 0.5) / height - 0.5
    xs = (np.arange(width, dtype=np.float64) + 0.5) / width - 0.5
    ay = np.abs(ys)[:, None]
    ax = np.abs(xs)[None, :]
    inside = (ax <= 0.41) & (ay <= 0.41) & (ax + ay <= 0.622)
    return inside.astype(np.float32)


def intrinsics_for(height, width):
    s = height / 256.0
    return np.array([[169.29275 * s, 0.0, 130.03175 * s],
                     [0.0, 169.29275 * s, 106.9795 * s],
                     [0.0, 0.0, 1.0]], dtype=np.float32)


def _rotation(rng, sigma):
    v = rng.normal(0.0, sigma, 3)
    angle = np.linalg.norm(v)
    if angle < 1e-12:
        return np.eye(3)
    k = v / angle
    kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(angle) * kx + (1 - np.cos(angle)) * (kx @ kx)


def make_batch(n, height, width, seed=0, sparse_points=500, gap_scale=None):
    """Returns {key: fp32 CPU tensor} for the 16 per-pair tensors of train.py:244-248.

    gap_scale: optional (lo, hi) -- per-sample frame gap g ~ U{lo..hi}, poses scaled by g / 10
    (BASELINE.json config 5, "adjacent range 5-30").
    """
    rng = np.random.default_rng(seed)
    mask = boundary_mask(height, width)
    inside = np.flatnonzero(mask.reshape(-1) > 0.5)
    sparse_points = min(sparse_points, inside.size)
    out = {}
    out["colors_1"] = rng.uniform(-1.0, 1.0, (n, 3, height, width)).astype(np.float32)
    out["colors_2"] = rng.uniform(-1.0, 1.0, (n, 3, height, width)).astype(np.float32)
    out["boundaries"] = np.broadcast_to(mask, (n, 1, height, width)).copy()
    out["intrinsics"] = np.broadcast_to(intrinsics_for(height, width), (n, 3, 3)).copy()
    r12 = np.zeros((n, 3, 3), np.float32)
    r21 = np.zeros((n, 3, 3), np.float32)
    t12 = np.zeros((n, 3, 1), np.float32)
    t21 = np.zeros((n, 3, 1), np.float32)
    for i in range(n):
        g = 1.0
        if gap_scale is not None:
            g = rng.integers(gap_scale[0], gap_scale[1] + 1) / 10.0
        r = _rotation(rng, 0.02 * g).astype(np.float32)
        t = (rng.normal(0.0, 0.03, (3, 1)) * g).astype(np.float32)
        r12[i], t12[i] = r, t
        r21[i] = r.T
        t21[i] = np.matmul(-r.T, t)
    out["rotations_1_wrt_2"], out["rotations_2_wrt_1"] = r12, r21
    out["translations_1_wrt_2"], out["translations_2_wrt_1"] = t12, t21
    for k in ("1", "2"):
        depth = np.zeros((n, height * width), np.float32)
        dmask = np.zeros((n, height * width), np.float32)
        flow = np.zeros((n, 2, height * width), np.float32)
        for i in range(n):
            loc = rng.choice(inside, sparse_points, replace=False)
            depth[i, loc] = rng.uniform(0.2, 0.8, sparse_points)
            dmask[i, loc] = 1.0
            flow[i, :, loc] = rng.normal(0.0, 0.03, (sparse_points, 2))
        out["sparse_depths_" + k] = depth.reshape(n, 1, height, width)
        out["sparse_depth_masks_" + k] = dmask.reshape(n, 1, height, width)
        out["sparse_flows_" + k] = flow.reshape(n, 2, height, width)
        out["sparse_flow_masks_" + k] = dmask.reshape(n, 1, height, width).copy()
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in out.items()}


def smooth_depth(n, height, width, seed=0, lo=0.3, hi=0.9):
    """A smooth positive depth-like field (sum of a few low-frequency cosines) for geometry tests."""
    rng = np.random.default_rng(seed)
    ys = np.linspace(0.0, 1.0, height)[None, :, None]
    xs = np.linspace(0.0, 1.0, width)[None, None, :]
    field = np.zeros((n, height, width))
    for _ in range(4):
        fy, fx = rng.uniform(0.5, 3.0, (2, n, 1, 1))
        ph = rng.uniform(0, 2 * np.pi, (n, 1, 1))
        field += np.cos(2 * np.pi * (fy * ys + fx * xs) + ph)
    field = (field - field.min()) / (field.max() - field.min() + 1e-12)
    field = lo + (hi - lo) * field + rng.normal(0.0, 0.01, field.shape)
    return torch.from_numpy(field.reshape(n, 1, height, width).astype(np.float32))
