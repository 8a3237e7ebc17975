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

    def __init__(self, point_cloud, mask_boundary, view_indexes_per_point, clean_point_list, visible_view_indexes, device="cuda",
                 extrinsics=None, projections=None, intrinsic_matrix=None, estimated_scale=None):
        """The first five arguments are those of reference utils.get_torch_training_data.  With the per-view ``extrinsics``
        (views x 4 x 4), ``projections`` (views x 3 x 4), the sequence's ``intrinsic_matrix`` and ``estimated_scale``
        (dataset.py:353-356, 386, 421) resident as well, ``training_batch`` assembles everything of a batch but the colour
        images on the device."""
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
        self.visible_view_indexes = [int(v) for v in visible_view_indexes]
        as64 = lambda a, shape: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))).to(self.device)
        self.extrinsics = None if extrinsics is None else as64(extrinsics, (-1, 4, 4))
        self.projections = None if projections is None else as64(projections, (-1, 3, 4))
        self.estimated_scale = None if estimated_scale is None else float(estimated_scale)
        self.intrinsics = None
        if intrinsic_matrix is not None:          # dataset.py:421-423
            k = np.asarray(intrinsic_matrix)[:3, :3].astype(np.float32).reshape(3, 3)
            self.intrinsics = torch.from_numpy(np.ascontiguousarray(k)).to(self.device)
        # dataset.py:427-430: the endoscope boundary as a {0, 1} plane
        boundary = self.mask.float() / 255.0
        self.boundary = (boundary > 0.9).float().reshape(1, 1, self.height, self.width)

    def planes(self, pair_extrinsics, pair_projections, pair_indexes, depth_multiplier=1.0):
        """pair_extrinsics (B,2,4,4), pair_projections (B,2,3,4), pair_indexes (B,2) frame indices.

        Returns dict of fp32 device tensors: ``depth_masks``, ``depths``, ``flow_masks`` (2,B,1,H,W) and
        ``flows`` (2,B,2,H,W); index 0 / 1 of the first axis = frame 1 / 2 of every pair.
        """
        to64 = lambda a, shape: (a.to(device=self.device, dtype=torch.float64) if torch.is_tensor(a)
                                 else torch.as_tensor(np.asarray(a, dtype=np.float64)).to(self.device)).reshape(shape).contiguous()
        ext = to64(pair_extrinsics, (-1, 2, 4, 4))
        proj = to64(pair_projections, (-1, 2, 3, 4))
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


    def relative_poses(self, pair_extrinsics):
        """dataset.py:384-399 on the device (endo_relative_poses): pair_extrinsics (B,2,4,4) fp64 device tensor ->
        rotations_1_wrt_2, rotations_2_wrt_1 (B,3,3), translations_1_wrt_2, translations_2_wrt_1 (B,3,1), fp32."""
        if self.estimated_scale is None:
            raise RuntimeError("SequenceScatter was built without estimated_scale")
        ext = pair_extrinsics.to(device=self.device, dtype=torch.float64).reshape(-1, 2, 4, 4).contiguous()
        batch = int(ext.shape[0])
        opts = dict(dtype=torch.float32, device=self.device)
        r12, r21 = torch.empty((batch, 3, 3), **opts), torch.empty((batch, 3, 3), **opts)
        t12, t21 = torch.empty((batch, 3, 1), **opts), torch.empty((batch, 3, 1), **opts)
        lib = _lib.load()
        with torch.cuda.device(self.device):
            rc = lib.endo_relative_poses(_lib.ptr(ext), batch, self.estimated_scale, _lib.ptr(r12), _lib.ptr(t12), _lib.ptr(r21),
                                         _lib.ptr(t21), _lib.stream())
        _lib.check(rc, "endo_relative_poses")
        return r12, r21, t12, t21

    def training_batch(self, positions):
        """Everything of a training batch except the two colour images, built on the device from the resident sequence
        (reference dataset.py:351-404, 419-430 for every sample of the batch): ``positions`` is a list of
        ``(pos, increment)`` as ``utils.generating_pos_and_increment`` returns them.  Returns the 14 tensors
        ``train_step.TrainingStep`` consumes besides ``colors_1`` / ``colors_2``, keyed as ``synthetic.BATCH_KEYS``; nothing is
        copied from the host but the position list."""
        if self.extrinsics is None or self.projections is None or self.intrinsics is None:
            raise RuntimeError("SequenceScatter was built without extrinsics / projections / intrinsic_matrix")
        first = torch.tensor([int(p) for p, _ in positions], dtype=torch.long, device=self.device)
        second = torch.tensor([int(p) + int(inc) for p, inc in positions], dtype=torch.long, device=self.device)
        batch = int(first.numel())
        pair_ext = torch.stack([self.extrinsics[first], self.extrinsics[second]], dim=1)           # (B,2,4,4)
        pair_proj = torch.stack([self.projections[first], self.projections[second]], dim=1)       # (B,2,3,4)
        pair_idx = [[self.visible_view_indexes[int(p)], self.visible_view_indexes[int(p) + int(inc)]] for p, inc in positions]
        planes = self.planes(pair_ext, pair_proj, pair_idx)
        scale = torch.tensor(self.estimated_scale, dtype=torch.float32, device=self.device)
        depths = planes["depths"] / scale          # dataset.py:391-392: float32 planes divided in place by the scale
        r12, r21, t12, t21 = self.relative_poses(pair_ext)
        return {"sparse_depths_1": depths[0], "sparse_depths_2": depths[1],
                "sparse_depth_masks_1": planes["depth_masks"][0], "sparse_depth_masks_2": planes["depth_masks"][1],
                "sparse_flows_1": planes["flows"][0], "sparse_flows_2": planes["flows"][1],
                "sparse_flow_masks_1": planes["flow_masks"][0], "sparse_flow_masks_2": planes["flow_masks"][1],
                "boundaries": self.boundary.expand(batch, 1, self.height, self.width).contiguous(),
                "rotations_1_wrt_2": r12, "rotations_2_wrt_1": r21, "translations_1_wrt_2": t12, "translations_2_wrt_1": t21,
                "intrinsics": self.intrinsics.reshape(1, 3, 3).expand(batch, 3, 3).contiguous()}


def get_torch_training_data(pair_extrinsics, pair_projections, pair_indexes, point_cloud, mask_boundary,
                            view_indexes_per_point, clean_point_list, visible_view_indexes, device="cuda"):
    """Drop-in for reference utils.py:460-612: returns (depth_masks, depths, flow_masks, flows) as float32 numpy
    arrays shaped (2,H,W,1), (2,H,W,1), (2,H,W,1), (2,H,W,2).  For training keep a ``SequenceScatter`` alive and
    use ``planes`` instead: this convenience form uploads the sequence on every call."""
    seq = SequenceScatter(point_cloud, mask_boundary, view_indexes_per_point, clean_point_list, visible_view_indexes, device)
    out = seq.planes(np.asarray(pair_extrinsics)[None], np.asarray(pair_projections)[None], np.asarray(pair_indexes)[None])
    to_hwc = lambda t: t[:, 0].permute(0, 2, 3, 1).contiguous().cpu().numpy()
    return to_hwc(out["depth_masks"]), to_hwc(out["depths"]), to_hwc(out["flow_masks"]), to_hwc(out["flows"])
