"""Sequence reader without OpenCV / plyfile / albumentations -- SURVEY.md 8 (f4).

Same function names, arguments and return values as the reference's reader functions (reference utils.py:29-36, 94-231,
232-285, 441-457, 72-83), so ``dataset.SfMDataset``-style code can call them unchanged:

  * the text formats of a sequence folder (``selected_indexes``, ``visible_view_indexes``, ``camera_intrinsics_per_view``,
    ``view_indexes_per_point``, ``motion.yaml``, ``structure.ply``) are parsed on the host, as the reference does;
  * the undistorted mask (``undistorted_mask.bmp``) is read, downsampled, eroded and cropped on the host once per sequence
    (``downsample_and_crop_mask``), with cv2's 8-bit INTER_LINEAR arithmetic restated -- bit-identical to the mask and crop
    window in the reference's own precompute file for the shipped example sequence;
  * the colour frames -- the per-iteration work -- go from the .jpg bytes to the cropped, downsampled image IN HBM through the
    C ABI (``endo_jpeg_decode_crop``, csrc/jpeg.hip): Huffman decoding on the calling thread into pinned memory, inverse DCT,
    chroma upsampling, colour conversion, cv2.resize and the crop on the GPU.  ``get_pair_color_imgs`` returns the uint8
    (2, H, W, 3) array of the reference as a device tensor; ``get_pair_color_tensors`` the normalised fp32 (2, 3, H, W) network
    input (dataset.py:148, 446-451), with no host image and no host-to-device image copy in between.

There is no CPU decoding path: without the HIP library the colour functions raise.
"""

import ctypes
import os
import struct

import numpy as np
import torch
import yaml

from . import _lib


# ---------------------------------------------------------------------------------------------
# text formats (utils.py:137-231)
# ---------------------------------------------------------------------------------------------
def _int_lines(path):
    with open(str(path)) as fp:
        return [int(line) for line in fp if line.strip()]


def read_selected_indexes(prefix_seq):
    """-> (stride, selected_indexes)   [utils.py:137-144]"""
    selected = _int_lines(os.path.join(str(prefix_seq), "selected_indexes"))
    return selected[1] - selected[0], selected


def read_visible_view_indexes(prefix_seq):
    """[utils.py:158-164]"""
    return _int_lines(os.path.join(str(prefix_seq), "visible_view_indexes"))


def read_visible_image_path_list(data_root):
    """Every index of every ``visible_view_indexes`` file below data_root  [utils.py:147-155]"""
    out = []
    for base, _, files in sorted(os.walk(str(data_root))):
        for name in sorted(files):
            if name.endswith("visible_view_indexes"):
                out.extend(_int_lines(os.path.join(base, name)))
    return out


def read_camera_intrinsic_per_view(prefix_seq):
    """fx, fy, cx, cy on four lines per view -> list of 3x4 float64 matrices  [utils.py:167-188]"""
    with open(os.path.join(str(prefix_seq), "camera_intrinsics_per_view")) as fp:
        numbers = [float(line) for line in fp if line.strip()]
    matrices = []
    for view in range(len(numbers) // 4):
        fx, fy, cx, cy = numbers[4 * view:4 * view + 4]
        matrices.append(np.array([[fx, 0.0, cx, 0.0], [0.0, fy, cy, 0.0], [0.0, 0.0, 1.0, 0.0]]))
    return matrices


def modify_camera_intrinsic_matrix(intrinsic_matrix, start_h, start_w, downsampling_factor):
    """Intrinsics of the downsampled, cropped image  [utils.py:191-198]"""
    out = np.array(intrinsic_matrix, dtype=np.float64, copy=True)
    out[0, 0] = intrinsic_matrix[0][0] / downsampling_factor
    out[1, 1] = intrinsic_matrix[1][1] / downsampling_factor
    out[0, 2] = intrinsic_matrix[0][2] / downsampling_factor - start_w
    out[1, 2] = intrinsic_matrix[1][2] / downsampling_factor - start_h
    return out


_PLY_TYPES = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h", "ushort": "H", "uint16": "H",
              "int": "i", "int32": "i", "uint": "I", "uint32": "I", "float": "f", "float32": "f", "double": "d", "float64": "d"}


def read_point_cloud(path):
    """Vertices of a PLY file as [x, y, z, ..., 1.0] lists, numbers in the file's property types  [utils.py:201-211]"""
    with open(str(path), "rb") as fp:
        raw = fp.read()
    marker = raw.find(b"end_header")
    if not raw.startswith(b"ply") or marker < 0:
        raise ValueError("%s is not a PLY file" % path)
    body_at = raw.index(b"\n", marker) + 1
    encoding, vertex_count, vertex_props, first_element = None, 0, [], None
    current = None
    for line in raw[:marker].decode("ascii").splitlines():
        words = line.split()
        if not words:
            continue
        if words[0] == "format":
            encoding = words[1]
        elif words[0] == "element":
            current = words[1]
            first_element = first_element or current
            if current == "vertex":
                vertex_count = int(words[2])
        elif words[0] == "property" and current == "vertex":
            if words[1] == "list":
                raise ValueError("list properties on vertices are not supported")
            vertex_props.append(words[1])
    if first_element != "vertex":
        raise ValueError("the vertex element must come first")
    dtypes = [np.dtype(_PLY_TYPES[t]) for t in vertex_props]
    points = []
    if encoding == "ascii":
        rows = raw[body_at:].decode("ascii").split("\n")[:vertex_count]
        for row in rows:
            fields = row.split()
            points.append([dt.type(f) for dt, f in zip(dtypes, fields)] + [1.0])
    else:
        order = "<" if encoding == "binary_little_endian" else ">"
        record = struct.Struct(order + "".join(_PLY_TYPES[t] for t in vertex_props))
        for n in range(vertex_count):
            fields = record.unpack_from(raw, body_at + n * record.size)
            points.append([dt.type(f) for dt, f in zip(dtypes, fields)] + [1.0])
    return points


def read_view_indexes_per_point(prefix_seq, visible_view_indexes, point_cloud_count):
    """(points, views) 0/1 float64 matrix: a negative line starts the next point  [utils.py:214-224]"""
    column = {}
    for i, view in enumerate(visible_view_indexes):
        column.setdefault(view, i)
    out = np.zeros((point_cloud_count, len(visible_view_indexes)))
    point = -1
    for value in _int_lines(os.path.join(str(prefix_seq), "view_indexes_per_point")):
        if value < 0:
            point += 1
        else:
            out[point, column[value]] = 1
    return out


def read_pose_data(prefix_seq):
    """The ``poses[]`` mapping of motion.yaml  [utils.py:225-231]"""
    with open(os.path.join(str(prefix_seq), "motion.yaml")) as stream:
        doc = yaml.safe_load(stream)
    return list(doc.values())[1]


def overlapping_visible_view_indexes_per_point(visible_view_indexes_per_point, visible_interval):
    """Window sums over the view axis, [i - interval, i + interval)  [utils.py:29-36]"""
    src = np.asarray(visible_view_indexes_per_point)
    count = src.shape[1]
    csum = np.concatenate([np.zeros((src.shape[0], 1)), np.cumsum(src, axis=1)], axis=1)
    lo = np.maximum(np.arange(count) - visible_interval, 0)
    hi = np.minimum(np.arange(count) + visible_interval, count)
    return csum[:, hi] - csum[:, lo]


def quaternion_matrix(quaternion):
    """Homogeneous rotation of (w, x, y, z)  [utils.py:1358-1382]"""
    q = np.array(quaternion, dtype=np.float64)
    n = float(q @ q)
    if n < np.finfo(float).eps * 4.0:
        return np.identity(4)
    q = q * np.sqrt(2.0 / n)
    o = np.outer(q, q)
    return np.array([[1.0 - o[2, 2] - o[3, 3], o[1, 2] - o[3, 0], o[1, 3] + o[2, 0], 0.0],
                     [o[1, 2] + o[3, 0], 1.0 - o[1, 1] - o[3, 3], o[2, 3] - o[1, 0], 0.0],
                     [o[1, 3] - o[2, 0], o[2, 3] + o[1, 0], 1.0 - o[1, 1] - o[2, 2], 0.0],
                     [0.0, 0.0, 0.0, 1.0]])


def get_extrinsic_matrix_and_projection_matrix(poses, intrinsic_matrix, visible_view_count):
    """World-to-camera 4x4 and 3x4 projection per visible view  [utils.py:264-285]"""
    extrinsics, projections = [], []
    for i in range(visible_view_count):
        pose = poses["poses[%d]" % i]
        rot, pos = pose["orientation"], pose["position"]
        camera_to_world = quaternion_matrix([rot["w"], rot["x"], rot["y"], rot["z"]])
        camera_to_world[:3, 3] = [pos["x"], pos["y"], pos["z"]]
        world_to_camera = np.linalg.inv(camera_to_world)
        extrinsics.append(world_to_camera)
        projections.append(np.dot(intrinsic_matrix, world_to_camera))
    return extrinsics, projections


def global_scale_estimation(extrinsics, point_cloud):
    """max(1, extent of the camera centres column, extent of the point cloud)  [utils.py:232-261]"""
    centres = np.stack([np.asarray(e)[:3, 3] for e in extrinsics]).reshape(len(extrinsics), 3)
    extent_cameras = np.linalg.norm(centres.max(axis=0) - centres.min(axis=0), ord=2)
    pts = np.asarray([p[:3] for p in point_cloud], dtype=np.float32)
    usable = ~np.isnan(pts).any(axis=1)
    usable[0] = True
    pts = pts[usable]
    extent_points = np.linalg.norm(pts.max(axis=0) - pts.min(axis=0), ord=2)
    return max(1.0, max(extent_cameras, extent_points))


# ---------------------------------------------------------------------------------------------
# mask (utils.py:94-135 and dataset.py:27, 47: cv2.imread(..., IMREAD_GRAYSCALE))
# ---------------------------------------------------------------------------------------------
def read_mask(path):
    """Uncompressed 8-bit paletted or 24-bit BMP as a grey uint8 image (cv2's BGR -> grey weights 1868 / 9617 / 4899, 14 bits)."""
    with open(str(path), "rb") as fp:
        raw = fp.read()
    if raw[:2] != b"BM":
        raise ValueError("%s is not a BMP file" % path)
    pixels_at, = struct.unpack_from("<I", raw, 10)
    header_size, width, height, _, bits, compression = struct.unpack_from("<IiiHHI", raw, 14)
    if compression != 0 or bits not in (8, 24):
        raise ValueError("only uncompressed 8 / 24 bit BMP masks are supported")
    bottom_up = height > 0
    height = abs(height)
    row_bytes = (width * bits + 31) // 32 * 4
    rows = np.frombuffer(raw, np.uint8, row_bytes * height, pixels_at).reshape(height, row_bytes)
    weights = np.array([1868, 9617, 4899], dtype=np.int64)
    if bits == 8:
        entries, = struct.unpack_from("<I", raw, 46)
        palette = np.frombuffer(raw, np.uint8, (entries or 256) * 4, 14 + header_size).reshape(-1, 4)[:, :3].astype(np.int64)
        grey = ((palette @ weights + 8192) >> 14).astype(np.uint8)
        image = grey[rows[:, :width]]
    else:
        bgr = rows[:, :3 * width].reshape(height, width, 3).astype(np.int64)
        image = ((bgr @ weights + 8192) >> 14).astype(np.uint8)
    return np.ascontiguousarray(image[::-1] if bottom_up else image)


def _taps(dst, src, scale):
    pos = ((np.arange(dst) + 0.5) * scale - 0.5).astype(np.float32)
    first = np.floor(pos).astype(np.int64)
    frac = pos - first.astype(np.float32)
    frac[first < 0] = 0.0
    first = np.maximum(first, 0)
    frac[first >= src - 1] = 0.0
    first = np.minimum(first, src - 1)
    second = np.minimum(first + 1, src - 1)
    return first, second, np.rint((1.0 - frac) * 2048.0).astype(np.int64), np.rint(frac * 2048.0).astype(np.int64)


def resize_u8(image, downsampling_factor):
    """cv2.resize(image, (0, 0), fx=1/d, fy=1/d) (INTER_LINEAR) for a grey uint8 image, in cv2's fixed-point arithmetic."""
    inv = 1.0 / downsampling_factor
    h, w = image.shape
    out_h, out_w = int(np.rint(h * inv)), int(np.rint(w * inv))
    x0, x1, ax0, ax1 = _taps(out_w, w, 1.0 / inv)
    y0, y1, ay0, ay1 = _taps(out_h, h, 1.0 / inv)
    src = image.astype(np.int64)
    horizontal = (src[:, x0] * ax0 + src[:, x1] * ax1) >> 4
    value = ((ay0[:, None] * horizontal[y0]) >> 16) + ((ay1[:, None] * horizontal[y1]) >> 16)
    return np.clip((value + 2) >> 2, 0, 255).astype(np.uint8)


def erode_u8(image, size=5):
    """cv2.erode(image, np.ones((size, size), np.uint8)): window minimum; positions outside the image are ignored."""
    r = size // 2
    h, w = image.shape
    framed = np.full((h + 2 * r, w + 2 * r), 255, np.uint8)
    framed[r:r + h, r:r + w] = image
    rows = framed[0:h]
    for k in range(1, size):
        rows = np.minimum(rows, framed[k:k + h])
    out = rows[:, 0:w]
    for k in range(1, size):
        out = np.minimum(out, rows[:, k:k + w])
    return out


def downsample_and_crop_mask(mask, downsampling_factor, divide, suggested_h=None, suggested_w=None):
    """-> (cropped eroded mask, start_h, end_h, start_w, end_w)   [utils.py:94-135]"""
    small = resize_u8(mask, downsampling_factor)
    limit_h, limit_w = small.shape
    ys, xs = np.nonzero(small == 255)
    span_h, span_w = int(ys.max() - ys.min()), int(xs.max() - xs.min())
    pad_h, pad_w = divide - span_h % divide, divide - span_w % divide
    start_h = max(int(ys.min()) - pad_h // 2, 0)
    start_w = max(int(xs.min()) - pad_w // 2, 0)
    end_h, end_w = start_h + span_h + pad_h, start_w + span_w + pad_w
    if suggested_h is not None and suggested_h != span_h:
        start_h = max(start_h - (suggested_h - (span_h + pad_h)) // 2, 0)
        end_h = min(suggested_h + start_h, limit_h)
        start_h = end_h - suggested_h
    if suggested_w is not None and suggested_w != span_w:
        start_w = max(start_w - (suggested_w - (span_w + pad_w)) // 2, 0)
        end_w = min(suggested_w + start_w, limit_w)
        start_w = end_w - suggested_w
    return erode_u8(small, 5)[start_h:end_h, start_w:end_w], start_h, end_h, start_w, end_w


# ---------------------------------------------------------------------------------------------
# colour frames (utils.py:441-457, 72-83; dataset.py:148, 446-451) -- on the device
# ---------------------------------------------------------------------------------------------
class FrameDecoder(object):
    """Staging (pinned host) and workspace (device) buffers of ``endo_jpeg_decode_crop``, grown on demand and reused.
    One slot per image that may be in flight: the staging buffer of slot s is rewritten only after the stream has passed
    the previous use of that slot (an event per slot)."""

    def __init__(self, device="cuda", slots=4):
        self.device = torch.device(device)
        self.slots = [dict(staging=None, workspace=None, event=None) for _ in range(slots)]
        self._next = 0

    def _slot(self, nbytes):
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        if slot["event"] is not None:
            slot["event"].synchronize()
        if slot["staging"] is None or slot["staging"].numel() < nbytes:
            slot["staging"] = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
            slot["workspace"] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return slot

    def decode(self, jpeg_bytes, start_h, end_h, start_w, end_w, downsampling_factor, rgb_mode="rgb", out_u8=None, out_f32=None):
        """Writes the crop of the downsampled frame into out_u8 (H, W, 3) uint8 and / or out_f32 (3, H, W) fp32 (device)."""
        lib = _lib.load()
        data = np.frombuffer(jpeg_bytes, dtype=np.uint8)
        src = ctypes.c_void_p(data.ctypes.data)
        need = int(lib.endo_jpeg_workspace_bytes(src, data.size))
        if need < 0:
            raise ValueError("not a JPEG file this reader supports (sequential Huffman, 8 bit, grey / 4:4:4 / 4:2:2 / 4:2:0)")
        slot = self._slot(need)
        h, w = end_h - start_h, end_w - start_w
        for t, shape, dt in ((out_u8, (h, w, 3), torch.uint8), (out_f32, (3, h, w), torch.float32)):
            if t is not None and (tuple(t.shape) != shape or t.dtype != dt or not t.is_cuda or not t.is_contiguous()):
                raise ValueError("output must be a contiguous device tensor of shape %s, %s" % (shape, dt))
        _lib.check(lib.endo_jpeg_decode_crop(src, data.size, float(downsampling_factor), int(start_h), int(end_h), int(start_w),
                                             int(end_w), 1 if rgb_mode == "rgb" else 0, _lib.ptr(out_u8), _lib.ptr(out_f32),
                                             ctypes.c_void_p(slot["staging"].data_ptr()), _lib.ptr(slot["workspace"]), need,
                                             _lib.stream()), "endo_jpeg_decode_crop")
        if slot["event"] is None:
            slot["event"] = torch.cuda.Event()
        slot["event"].record()


_default_decoder = None


def _decoder():
    global _default_decoder
    if _default_decoder is None:
        _default_decoder = FrameDecoder()
    return _default_decoder


def _frame_bytes(prefix_seq, index):
    with open(os.path.join(str(prefix_seq), "%08d.jpg" % index), "rb") as fp:
        return fp.read()


def get_pair_color_imgs(prefix_seq, pair_indexes, start_h, end_h, start_w, end_w, downsampling_factor, is_hsv=False, rgb_mode="rgb",
                        decoder=None):
    """uint8 (len(pair_indexes), H, W, 3) DEVICE tensor with the values of the reference's numpy array  [utils.py:441-457].
    is_hsv (train.py --use_hsv_colorspace): the BGR frame through cv2.COLOR_BGR2HSV_FULL (endo_hsv_full: OpenCV's 8-bit fixed-point
    arithmetic, parity unpinned against cv2 itself); rgb_mode is then not looked at, as in the reference."""
    decoder = decoder or _decoder()
    out = torch.empty((len(pair_indexes), end_h - start_h, end_w - start_w, 3), dtype=torch.uint8, device=decoder.device)
    for k, index in enumerate(pair_indexes):
        decoder.decode(_frame_bytes(prefix_seq, index), start_h, end_h, start_w, end_w, downsampling_factor, "bgr" if is_hsv else rgb_mode, out_u8=out[k])
    if is_hsv:
        hsv_full(out, blue_index=0, out_u8=out)
    return out


def hsv_full(img_u8, blue_index=0, out_u8=None, out_f32=None):
    """cv2.cvtColor(img, COLOR_BGR2HSV_FULL) (blue_index 0) / COLOR_RGB2HSV_FULL (2) of a uint8 (..., H, W, 3) device tensor into a uint8
    tensor of the same shape and / or, for ONE image, the Normalize(0.5, 0.5) fp32 (3, H, W) tensor of dataset.py:446-451; in place when
    out_u8 is img_u8  [utils.py:449-450, 80-81; dataset.py:439-442]."""
    lib = _lib.load()
    if img_u8.dtype != torch.uint8 or not img_u8.is_cuda or not img_u8.is_contiguous() or img_u8.shape[-1] != 3:
        raise ValueError("expected a contiguous uint8 device tensor (..., H, W, 3)")
    pixels = img_u8.numel() // 3
    if out_u8 is None and out_f32 is None:
        out_u8 = torch.empty_like(img_u8)
    if out_f32 is not None and (img_u8.dim() != 3 or tuple(out_f32.shape) != (3, img_u8.shape[0], img_u8.shape[1]) or out_f32.dtype != torch.float32
                                or not out_f32.is_contiguous()):
        raise ValueError("out_f32 is the (3, H, W) fp32 tensor of one (H, W, 3) image")
    _lib.check(lib.endo_hsv_full(_lib.ptr(img_u8), pixels, int(blue_index), _lib.ptr(out_u8), _lib.ptr(out_f32), _lib.stream()), "endo_hsv_full")
    return out_u8 if out_u8 is not None else out_f32


def get_pair_color_tensors(prefix_seq, pair_indexes, start_h, end_h, start_w, end_w, downsampling_factor, rgb_mode="rgb", decoder=None, is_hsv=False):
    """fp32 (len(pair_indexes), 3, H, W) device tensor: get_pair_color_imgs + Normalize(0.5, 0.5) + img_to_tensor, the network's
    colour input when no augmentation runs in between (dataset.py:446-451 validation branch; evaluate.py)."""
    decoder = decoder or _decoder()
    h, w = end_h - start_h, end_w - start_w
    out = torch.empty((len(pair_indexes), 3, h, w), dtype=torch.float32, device=decoder.device)
    scratch = torch.empty((h, w, 3), dtype=torch.uint8, device=decoder.device) if is_hsv else None
    for k, index in enumerate(pair_indexes):
        if is_hsv:
            decoder.decode(_frame_bytes(prefix_seq, index), start_h, end_h, start_w, end_w, downsampling_factor, "bgr", out_u8=scratch)
            hsv_full(scratch, blue_index=0, out_f32=out[k])
        else:
            decoder.decode(_frame_bytes(prefix_seq, index), start_h, end_h, start_w, end_w, downsampling_factor, rgb_mode, out_f32=out[k])
    return out


def get_test_color_img(img_file_name, start_h, end_h, start_w, end_w, downsampling_factor, is_hsv=False, rgb_mode="rgb", decoder=None):
    """fp32 (H, W, 3) device tensor holding the uint8 values  [utils.py:72-83]; is_hsv: cv2.COLOR_BGR2HSV_FULL of the BGR frame"""
    decoder = decoder or _decoder()
    out = torch.empty((end_h - start_h, end_w - start_w, 3), dtype=torch.uint8, device=decoder.device)
    with open(str(img_file_name), "rb") as fp:
        decoder.decode(fp.read(), start_h, end_h, start_w, end_w, downsampling_factor, "bgr" if is_hsv else rgb_mode, out_u8=out)
    if is_hsv:
        hsv_full(out, blue_index=0, out_u8=out)
    return out.float()


def get_color_imgs(prefix_seq, visible_view_indexes, start_h, end_h, start_w, end_w, downsampling_factor, is_hsv=False, decoder=None):
    """uint8 (views, H, W, 3) DEVICE tensor in cv2 order (B, G, R): the values of the reference's float32 array  [utils.py:288-300]"""
    return get_pair_color_imgs(prefix_seq, visible_view_indexes, start_h, end_h, start_w, end_w, downsampling_factor, is_hsv, "bgr", decoder)          # is_hsv: HSV_FULL frames, as the reference returns them


# ---------------------------------------------------------------------------------------------
# contaminated-point filter (utils.py:303-404)
# ---------------------------------------------------------------------------------------------
def compute_sanity_threshold(sanity_array, inlier_percentage):
    """Edges of the window around the histogram's peak bin that holds `inlier_percentage` of the samples  [utils.py:303-336]"""
    edges = np.arange(1000) * np.max(sanity_array) / 1000.0
    counts, edges = np.histogram(sanity_array, bins=edges, density=True)
    share = counts * np.diff(edges)
    peak = int(np.argmax(share))
    covered, above, below = share[peak], 1, 1
    while True:
        if peak + above < len(share):
            covered += share[peak + above]
            above += 1
            if covered >= inlier_percentage:
                break
        if peak - below >= 0:
            covered += share[peak - below]
            below += 1
            if covered >= inlier_percentage:
                break
        if peak + above >= len(share) and peak - below < 0:
            return np.min(edges), np.max(edges)
    return edges[peak - below + 1], edges[peak + above]


def point_brightness(imgs, point_cloud, view_indexes_per_point, mask_boundary, projection_matrices, extrinsic_matrices,
                     d=7, sigma_color=25.0, sigma_space=25.0):
    """Per frame and point: (valid (F, P) bool, camera depth (F, P) float64, filtered brightness (F, P) float32) as numpy arrays.
    imgs: device uint8 (F, H, W, 3) in cv2 order (reader.get_color_imgs).  One kernel evaluates the bilateral filter only at the
    pixels the points project to (endo_point_brightness)."""
    lib = _lib.load()
    if not (torch.is_tensor(imgs) and imgs.is_cuda and imgs.dtype == torch.uint8 and imgs.dim() == 4 and imgs.shape[3] == 3):
        raise ValueError("imgs must be a device uint8 tensor (frames, H, W, 3)")
    imgs = imgs.contiguous()
    frames, height, width, _ = imgs.shape
    dev = imgs.device
    pts = torch.from_numpy(np.ascontiguousarray(np.asarray(point_cloud, dtype=np.float64).reshape(-1, 4))).to(dev)
    n_points = int(pts.shape[0])
    proj = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(m, dtype=np.float64).reshape(3, 4) for m in projection_matrices]))).to(dev)
    ext = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(m, dtype=np.float64).reshape(4, 4) for m in extrinsic_matrices]))).to(dev)
    if proj.shape[0] != frames or ext.shape[0] != frames:
        raise ValueError("one projection and one extrinsic matrix per frame")
    vis = torch.from_numpy(np.ascontiguousarray(np.asarray(view_indexes_per_point, dtype=np.float32).reshape(n_points, frames))).to(dev)
    mask = torch.from_numpy(np.ascontiguousarray(np.asarray(mask_boundary, dtype=np.uint8).reshape(height, width))).to(dev)
    valid = torch.empty((frames, n_points), dtype=torch.int32, device=dev)
    depth = torch.empty((frames, n_points), dtype=torch.float64, device=dev)
    bright = torch.empty((frames, n_points), dtype=torch.float32, device=dev)
    _lib.check(lib.endo_point_brightness(_lib.ptr(imgs), frames, height, width, _lib.ptr(pts), n_points, _lib.ptr(proj), _lib.ptr(ext),
                                         _lib.ptr(vis), _lib.ptr(mask), int(d), float(sigma_color), float(sigma_space), _lib.ptr(valid),
                                         _lib.ptr(depth), _lib.ptr(bright), _lib.stream()), "endo_point_brightness")
    return valid.cpu().numpy() != 0, depth.cpu().numpy(), bright.cpu().numpy()


def get_clean_point_list(imgs, point_cloud, view_indexes_per_point, mask_boundary, inlier_percentage, projection_matrices,
                         extrinsic_matrices, is_hsv=False):
    """float32 (points,) 1 = keep, 0 = contaminated: in at least half of the frames it appears in, the point's depth^2 x brightness
    lies outside the window holding `inlier_percentage` of that frame's points  [utils.py:339-404].  imgs as reader.get_color_imgs in
    cv2 order (B, G, R).  The reference's is_hsv branch first takes its HSV frames BACK to BGR (cv2.COLOR_HSV2BGR_FULL, utils.py:362-363)
    and then runs the same filter.  That inverse conversion is not built here, and the kernel below reads its input as B, G, R: frames
    from get_color_imgs(..., is_hsv=True) would be filtered as if H, S, V were colours.  So is_hsv=True raises; hand over the BGR
    frames (get_color_imgs(..., is_hsv=False), as reader.load_sequence does) -- the filter then sees what the reference's sees, up to
    the 8-bit loss of its round trip."""
    if is_hsv:
        raise NotImplementedError("get_clean_point_list filters BGR frames: decode them with get_color_imgs(..., is_hsv=False); "
                                  "the reference's HSV -> BGR round trip (utils.py:362-363) is not implemented")
    n_points = len(point_cloud)
    if inlier_percentage <= 0.0 or inlier_percentage >= 1.0:
        return list()
    valid, depth, bright = point_brightness(imgs, point_cloud, view_indexes_per_point, mask_boundary, projection_matrices, extrinsic_matrices)
    flagged = np.zeros(n_points, dtype=np.int32)
    seen = np.zeros(n_points, dtype=np.int32)
    for f in range(valid.shape[0]):
        index = np.nonzero(valid[f])[0]
        seen[index] += 1
        if index.size < 2:
            continue
        sanity = depth[f, index] ** 2 * bright[f, index]
        low, high = compute_sanity_threshold(sanity, inlier_percentage)
        flagged[index[(sanity <= low) | (sanity >= high)]] += 1
    return (flagged < seen / 2).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# one sequence folder -> what dataset.pre_processing_data collects (dataset.py:41-112), minus the contaminated-point filter
# ---------------------------------------------------------------------------------------------
def load_sequence(folder, downsampling, network_downsampling, visible_interval, suggested_h=None, suggested_w=None, inlier_percentage=None):
    """Dictionary with the per-sequence entries of the reference's precompute file: crop_positions, selected_indexes,
    visible_view_indexes, point_cloud, intrinsic_matrix, mask_boundary, view_indexes_per_point, extrinsics, projection,
    estimated_scale and -- with inlier_percentage (train.py --inlier_percentage, 0.99) -- clean_point_list: every visible frame
    decoded on the device and the contaminated-point filter run on them (dataset.py:96-111).  This part needs the GPU."""
    folder = str(folder)
    mask, start_h, end_h, start_w, end_w = downsample_and_crop_mask(read_mask(os.path.join(folder, "undistorted_mask.bmp")),
                                                                    downsampling, network_downsampling, suggested_h, suggested_w)
    _, selected = read_selected_indexes(folder)
    visible = read_visible_view_indexes(folder)
    intrinsics = modify_camera_intrinsic_matrix(read_camera_intrinsic_per_view(folder)[0], start_h, start_w, downsampling)
    points = read_point_cloud(os.path.join(folder, "structure.ply"))
    views = overlapping_visible_view_indexes_per_point(read_view_indexes_per_point(folder, visible, len(points)), visible_interval)
    extrinsics, projections = get_extrinsic_matrix_and_projection_matrix(read_pose_data(folder), intrinsics, len(visible))
    clean = None
    if inlier_percentage is not None:
        frames = get_color_imgs(folder, visible, start_h, end_h, start_w, end_w, downsampling)
        clean = get_clean_point_list(frames, points, views, mask, inlier_percentage, projections, extrinsics)
    return {"clean_point_list": clean, "crop_positions": [start_h, end_h, start_w, end_w], "selected_indexes": selected, "visible_view_indexes": visible,
            "point_cloud": points, "intrinsic_matrix": intrinsics, "mask_boundary": mask, "view_indexes_per_point": views,
            "extrinsics": extrinsics, "projection": projections, "estimated_scale": global_scale_estimation(extrinsics, points)}
