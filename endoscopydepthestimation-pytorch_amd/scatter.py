"""Sparse SfM scatter on the device -- the GPU side of reference ``utils.get_torch_training_data``
(utils.py:460-612) and of the per-sample assembly in ``dataset.py:384-404``.

``SequenceScatter`` keeps one sequence's point cloud, visibility matrix, clean-point flags and endoscope mask
resident in HBM; ``planes`` turns a batch of frame pairs (their extrinsic / projection matrices) into the
sparse depth / flow / mask planes of the training step, already NCHW on the device.
``get_torch_training_data`` is the reference function's drop-in (same arguments, same four arrays).
There is no CPU fallback: everything goes through ``endo_sparse_scatter`` (include/endo_hip.h).
"""

import numpy as np
import torch

from . import _lib


class SequenceScatter:
    """Device-resident SfM data of one sequence (reference utils.py:234-409 loads these per sequence)."""

    def __init__(self, point_cloud, mask_boundary, view_indexes_per_point, clean_point_list, visible_view_indexes, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SequenceScatter needs a GPU device: the MI355X path has no CPU fallback")
        _lib.load()
        mask = np.asarray(mask_boundary)
        self.height, self.width = int(mask.shape[0]), int(mask.shape[1])
        points = np.ascontiguousarray(np.asarray(point_cloud, dtype=np.float64).reshape(-1, 4))
        self.n_points = int(points.shape[0])
        self.points = torch.from_numpy(points).to(self.device)
        self.mask = torch.from_numpy(np.ascontiguousarray(mask.reshape(self.height, self.width)).astype(np.uint8)).to(self.device)
        vis = np.asarray(view_indexes_per_point, dtype=np.float32).reshape(self.n_points, len(visible_view_indexes))
        self.visibility = torch.from_numpy(np.ascontiguousarray(vis)).to(self.device)          # (P, views)
        clean = np.asarray(clean_point_list, dtype=np.float32).reshape(-1)
        self.clean = torch.from_numpy(clean).to(self.device) if clean.size else None           # utils.py:496: empty list = no filter
        self.view_column = {int(v): i for i, v in enumerate(visible_view_indexes)}

    def planes(self, pair_extrinsics, pair_projections, pair_indexes, depth_multiplier=1.0):
        """pair_extrinsics (B,2,4,4), pair_projections (B,2,3,4), pair_indexes (B,2) frame indices.

        Returns dict of fp32 device tensors: ``depth_masks``, ``depths``, ``flow_masks`` (2,B,1,H,W) and
        ``flows`` (2,B,2,H,W); index 0 / 1 of the first axis = frame 1 / 2 of every pair.
        """
        ext = torch.as_tensor(np.asarray(pair_extrinsics, dtype=np.float64)).reshape(-1, 2, 4, 4).contiguous().to(self.device)
        proj = torch.as_tensor(np.asarray(pair_projections, dtype=np.float64)).reshape(-1, 2, 3, 4).contiguous().to(self.device)
        batch = int(ext.shape[0])
        if int(proj.shape[0]) != batch:
            raise ValueError("pair_extrinsics and pair_projections disagree on the batch size")
        idx = np.asarray(pair_indexes).reshape(batch, 2)
        cols = torch.tensor([[self.view_column[int(v)] for v in row] for row in idx], dtype=torch.long, device=self.device)
        vis = self.visibility.t()[cols]                       # (B, 2, P)
        vis = vis.permute(0, 2, 1).contiguous()               # (B, P, 2)
        h, w = self.height, self.width
        opts = dict(dtype=torch.float32, device=self.device)
        out = {"depth_masks": torch.empty((2, batch, 1, h, w), **opts), "depths": torch.empty((2, batch, 1, h, w), **opts),
               "flow_masks": torch.empty((2, batch, 1, h, w), **opts), "flows": torch.empty((2, batch, 2, h, w), **opts)}
        winner = torch.empty((2, batch, h, w), dtype=torch.int32, device=self.device)
        lib = _lib.load()
        with torch.cuda.device(self.device):
            rc = lib.endo_sparse_scatter(_lib.ptr(self.points), self.n_points, _lib.ptr(proj), _lib.ptr(ext), _lib.ptr(vis),
                                         _lib.ptr(self.clean), _lib.ptr(self.mask), batch, h, w, float(depth_multiplier),
                                         _lib.ptr(winner), _lib.ptr(out["depth_masks"]), _lib.ptr(out["depths"]),
                                         _lib.ptr(out["flow_masks"]), _lib.ptr(out["flows"]), _lib.stream())
        _lib.check(rc, "endo_sparse_scatter")
        return out


def get_torch_training_data(pair_extrinsics, pair_projections, pair_indexes, point_cloud, mask_boundary,
                            view_indexes_per_point, clean_point_list, visible_view_indexes, device="cuda"):
    """Drop-in for reference utils.py:460-612: returns (depth_masks, depths, flow_masks, flows) as float32 numpy
    arrays shaped (2,H,W,1), (2,H,W,1), (2,H,W,1), (2,H,W,2).  For training keep a ``SequenceScatter`` alive and
    use ``planes`` instead: this convenience form uploads the sequence on every call."""
    seq = SequenceScatter(point_cloud, mask_boundary, view_indexes_per_point, clean_point_list, visible_view_indexes, device)
    out = seq.planes(np.asarray(pair_extrinsics)[None], np.asarray(pair_projections)[None], np.asarray(pair_indexes)[None])
    to_hwc = lambda t: t[:, 0].permute(0, 2, 3, 1).contiguous().cpu().numpy()
    return to_hwc(out["depth_masks"]), to_hwc(out["depths"]), to_hwc(out["flow_masks"]), to_hwc(out["flows"])
