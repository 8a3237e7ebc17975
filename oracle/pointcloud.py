"""Oracle (test infrastructure): coloured point cloud of a depth map, numpy float32 exactly as the reference.

Restates reference utils.py:823-852 (point_cloud_from_depth), the back-projection evaluate.py:272,340 call per frame:
every ``downsampling``-th pixel inside the mask becomes (x, y, z, r, g, b) with x = (w - cx) / fx * z,
y = (h - cy) / fy * z, in row-major pixel order.  dtype semantics are this container's numpy 2 (NEP 50) with the
float32 intrinsics / depth evaluate.py passes: a Python int minus a float32 scalar is float32, so every operation
is a single float32 rounding (pinned by tests/golden/point_cloud.npz, generated from the reference itself).
"""

import numpy as np


def point_cloud_from_depth(depth_map, color_img, mask_img, intrinsic_matrix, point_cloud_downsampling,
                           min_threshold=None, max_threshold=None):
    depth = np.asarray(depth_map, dtype=np.float32)
    height, width = depth.shape
    k = np.asarray(intrinsic_matrix, dtype=np.float32)
    fx, cx, fy, cy = k[0, 0], k[0, 2], k[1, 1], k[1, 2]
    hh, ww = np.mgrid[0:height, 0:width]
    keep = (hh % point_cloud_downsampling == 0) & (ww % point_cloud_downsampling == 0) & (np.asarray(mask_img) > 0.5)
    color = np.asarray(color_img)
    b, g, r = color[..., 0], color[..., 1], color[..., 2]
    if max_threshold is not None and min_threshold is not None:
        keep &= (np.maximum(np.maximum(r, g), b) >= max_threshold) & (np.minimum(np.minimum(r, g), b) <= min_threshold)
    x = (ww.astype(np.float32) - cx) / fx * depth
    y = (hh.astype(np.float32) - cy) / fy * depth
    cols = [x, y, depth, r.astype(np.uint8).astype(np.float32), g.astype(np.uint8).astype(np.float32),
            b.astype(np.uint8).astype(np.float32)]
    return np.stack([c[keep] for c in cols], axis=1).astype(np.float32).reshape(-1, 6)
