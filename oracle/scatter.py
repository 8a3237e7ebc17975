"""Oracle (test infrastructure): sparse SfM scatter, numpy float64/float32 exactly as the reference.

Restates reference utils.py:460-612 (get_torch_training_data): project the SfM point cloud into
both frames of a pair, keep points that are visible, clean, inside the image, in front of the
camera and on the endoscope mask, and scatter depth / flow / masks into H x W planes.  Pixel
collisions resolve as numpy fancy-index assignment does: the highest point index wins.
"""

import numpy as np


def _project(points, projection, extrinsic):
    uvw = points @ projection.T
    uv = np.round(uvw / uvw[:, 2:3])
    cam = points @ extrinsic.T
    cam = cam / cam[:, 3:4]
    return uv, cam


def sparse_planes(pair_extrinsics, pair_projections, visibility_pair, clean_points, points, mask):
    """visibility_pair: (P, 2) visibility flags of the two frames; clean_points: (P,) or empty.

    Returns (depth_masks, depths, flow_masks, flows) shaped (2,H,W,1), (2,H,W,1), (2,H,W,1),
    (2,H,W,2) float32, i.e. the four arrays of utils.py:612.
    """
    height, width = mask.shape[:2]
    points = np.asarray(points).reshape(-1, 4)
    flat_mask = mask.reshape(-1)
    uv, cam = [], []
    for i in range(2):
        a, b = _project(points, pair_projections[i], pair_extrinsics[i])
        uv.append(a)
        cam.append(b)

    depth_masks = np.zeros((2, height * width, 1), np.float32)
    depths = np.zeros((2, height * width, 1), np.float32)
    flow_masks = np.zeros((2, height * width, 1), np.float32)
    flows = np.zeros((2, height * width, 2), np.float32)
    for i in range(2):
        keep = visibility_pair[:, i] > 0.5
        if len(clean_points) != 0:
            keep = keep & (np.asarray(clean_points).reshape(-1) > 0.5)
        keep = keep & (uv[i][:, 0] <= width - 1) & (uv[i][:, 0] >= 0)
        keep = keep & (uv[i][:, 1] <= height - 1) & (uv[i][:, 1] >= 0) & (cam[i][:, 2] > 0)
        idx = np.nonzero(keep)[0]
        loc = (np.round(uv[i][idx, 0]) + np.round(uv[i][idx, 1]) * width).astype(np.int32)
        on_mask = flat_mask[loc] == 255
        idx, loc = idx[on_mask], loc[on_mask]
        flow_masks[i, loc, 0] = 1.0
        flows[i, loc, :] = uv[1 - i][idx, :2] - uv[i][idx, :2]
        flows[i, :, 0] /= width
        flows[i, :, 1] /= height
        outlier = (np.abs(flows[i, :, 0]) > 5.0) | (np.abs(flows[i, :, 1]) > 5.0)
        flow_masks[i, outlier, 0] = 0.0
        flows[i, outlier, :] = 0.0
        depths[i, loc, 0] = cam[i][idx, 2]
        depth_masks[i, loc, 0] = 1.0
    shape = (2, height, width)
    return (depth_masks.reshape(shape + (1,)), depths.reshape(shape + (1,)),
            flow_masks.reshape(shape + (1,)), flows.reshape(shape + (2,)))
