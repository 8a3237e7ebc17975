"""Oracle (test infrastructure): relative camera motion of a frame pair and the rest of the per-sample host assembly,
numpy, restating reference dataset.py:384-399, 427-430 line by line.

These lines live inside ``SfMDataset.__getitem__``, between an image read that needs OpenCV and an augmentation that needs
albumentations, so the reference function cannot be run here and there is no reference-produced fixture for them: parity
unpinned for this helper beyond the identities checked in tests (R_2wrt1 R_1wrt2 = I, t round trip) -- it is eight lines of
numpy, restated verbatim.
"""
import numpy as np


def relative_poses(extrinsic_1, extrinsic_2, estimated_scale):
    """dataset.py:384-399.  Returns (rotation_1_wrt_2 (3,3), rotation_2_wrt_1 (3,3), translation_1_wrt_2 (3,1),
    translation_2_wrt_1 (3,1)), all float32."""
    relative_motion = np.matmul(extrinsic_1, np.linalg.inv(extrinsic_2))
    rotation_1_wrt_2 = np.reshape(relative_motion[:3, :3], (3, 3)).astype(np.float32)
    translation_1_wrt_2 = (np.reshape(relative_motion[:3, 3], (3, 1)) / estimated_scale).astype(np.float32)
    rotation_2_wrt_1 = np.transpose(rotation_1_wrt_2).astype(np.float32)
    translation_2_wrt_1 = np.matmul(-np.transpose(rotation_1_wrt_2), translation_1_wrt_2).astype(np.float32)
    return rotation_1_wrt_2, rotation_2_wrt_1, translation_1_wrt_2.reshape((3, 1)), translation_2_wrt_1.reshape((3, 1))


def boundary_plane(mask_boundary):
    """dataset.py:427-430: uint8 endoscope mask -> float32 {0, 1} plane (H, W)."""
    mask = np.asarray(mask_boundary).astype(np.float32) / 255.0
    mask[mask > 0.9] = 1.0
    mask[mask <= 0.9] = 0.0
    return mask
