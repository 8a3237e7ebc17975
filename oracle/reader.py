"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's sequence reader, SURVEY.md section 8 row (f4).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What it restates (reference file:line):
  * utils.py:137-231   read_selected_indexes, read_visible_view_indexes, read_camera_intrinsic_per_view,
                       modify_camera_intrinsic_matrix, read_point_cloud, read_view_indexes_per_point, read_pose_data
  * utils.py:29-36     overlapping_visible_view_indexes_per_point
  * utils.py:94-135    downsample_and_crop_mask (cv2.resize + cv2.erode + crop)
  * utils.py:232-261   global_scale_estimation
  * utils.py:264-285   get_extrinsic_matrix_and_projection_matrix (+ quaternion_matrix, utils.py:1358-1382)
  * utils.py:441-457   get_pair_color_imgs (cv2.imread + cv2.resize + crop + BGR2RGB)

Third-party pieces the reference calls and this image lacks, restated from their published algorithms:
  * OpenCV (cv2, version not pinned by the reference: no requirements file) --
      cv2.imread(jpg): libjpeg(-turbo) with its defaults: JDCT_ISLOW inverse DCT (jidctint.c), "fancy" h2v2 chroma
        upsampling (jdsample.c h2v2_fancy_upsample) and the integer YCbCr->RGB tables (jdcolor.c).  `decode_jpeg_pil` decodes
        with THE library (Pillow's bundled libjpeg-turbo); `decode_jpeg_blocks` is the numpy restatement of those three steps
        from entropy-decoded coefficient blocks (what the HIP kernels implement), pinned here against the library's output.
      cv2.resize(src, (0, 0), fx, fy) with the default INTER_LINEAR on uint8: source coordinate (d + 0.5) / f - 0.5, 11-bit
        coefficients, two-pass fixed point `(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2` (imgproc/resize.cpp,
        HResizeLinear / VResizeLinear<uchar, int, short>).
      cv2.erode(5x5 ones): minimum over the window, outside pixels ignored (BORDER_CONSTANT with the morphology default +max).
      cv2.imread(bmp, IMREAD_GRAYSCALE): palette entry -> (B*1868 + G*9617 + R*4899 + 8192) >> 14.
    PARITY PINNING: the mask path (resize + erode + crop positions) and every text reader are checked against the reference's
    own precompute pickle for the shipped example sequence (tests/golden/reader_example.npz, written by the real reference
    with the real cv2).  The colour path has no cv2-produced vector in the reference tree: it is pinned against libjpeg-turbo
    through Pillow and the resize arithmetic through the mask; against cv2.imread itself it is "parity unpinned".
  * plyfile.PlyData (read_point_cloud): ASCII PLY 1.0 vertex element, restated from the format; pinned against the pickle's
    point cloud.
"""

import os
import struct

import numpy as np
import yaml


# ---------------------------------------------------------------------------------------------
# text readers -- utils.py:137-231
# ---------------------------------------------------------------------------------------------
def read_selected_indexes(prefix_seq):
    with open(os.path.join(str(prefix_seq), "selected_indexes")) as fp:
        selected = [int(line) for line in fp]
    return selected[1] - selected[0], selected


def read_visible_view_indexes(prefix_seq):
    with open(os.path.join(str(prefix_seq), "visible_view_indexes")) as fp:
        return [int(line) for line in fp]


def read_camera_intrinsic_per_view(prefix_seq):
    """Four lines per view: fx, fy, cx, cy -> 3x4 matrices (utils.py:167-188)."""
    out = []
    with open(os.path.join(str(prefix_seq), "camera_intrinsics_per_view")) as fp:
        values = [float(line) for line in fp if line.strip()]
    for i in range(0, len(values) - 3, 4):
        m = np.zeros((3, 4))
        m[0, 0], m[1, 1], m[0, 2], m[1, 2], m[2, 2] = values[i], values[i + 1], values[i + 2], values[i + 3], 1.0
        out.append(m)
    return out


def modify_camera_intrinsic_matrix(intrinsic_matrix, start_h, start_w, downsampling_factor):
    m = np.copy(intrinsic_matrix)
    m[0][0] = intrinsic_matrix[0][0] / downsampling_factor
    m[1][1] = intrinsic_matrix[1][1] / downsampling_factor
    m[0][2] = intrinsic_matrix[0][2] / downsampling_factor - start_w
    m[1][2] = intrinsic_matrix[1][2] / downsampling_factor - start_h
    return m


def read_point_cloud(path):
    """[x, y, z, 1.0] per vertex; x, y, z as float32 (the PLY property type), as plyfile returns them (utils.py:201-211)."""
    with open(str(path), "rb") as fp:
        raw = fp.read()
    end = raw.index(b"end_header") + len(b"end_header")
    header = raw[:end].decode("ascii").split("\n")
    body = raw[end:].lstrip(b"\r").lstrip(b"\n")
    fmt = [l.split()[1] for l in header if l.startswith("format")][0]
    elements = []          # (name, count, [(type, name)])
    for line in header:
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "element":
            elements.append([tok[1], int(tok[2]), []])
        elif tok[0] == "property":
            elements[-1][2].append((tok[1], tok[-1]))
    name, count, props = elements[0]
    assert name == "vertex"
    points = []
    if fmt == "ascii":
        lines = body.decode("ascii").split("\n")
        for n in range(count):
            vals = lines[n].split()
            points.append([np.float32(v) for v in vals[:len(props)]] + [1.0])
    else:
        code = {"float": "f", "float32": "f", "double": "d", "uchar": "B", "uint8": "B", "int": "i", "int32": "i"}
        rec = struct.Struct(("<" if "little" in fmt else ">") + "".join(code[t] for t, _ in props))
        for n in range(count):
            vals = rec.unpack_from(body, n * rec.size)
            points.append([np.float32(v) if t.startswith("float") else v for v, (t, _) in zip(vals, props)] + [1.0])
    return points


def read_view_indexes_per_point(prefix_seq, visible_view_indexes, point_cloud_count):
    out = np.zeros((point_cloud_count, len(visible_view_indexes)))
    position = {v: i for i, v in reversed(list(enumerate(visible_view_indexes)))}          # list.index: first occurrence
    point = -1
    with open(os.path.join(str(prefix_seq), "view_indexes_per_point")) as fp:
        for line in fp:
            v = int(line)
            if v < 0:
                point += 1
            else:
                out[point][position[v]] = 1
    return out


def read_pose_data(prefix_seq):
    """motion.yaml -> the `poses[]` mapping (utils.py:225-231: `keys, values = doc.items(); poses = values[1]`)."""
    with open(os.path.join(str(prefix_seq), "motion.yaml")) as stream:
        doc = yaml.safe_load(stream)
    (_, _), (_, poses) = doc.items()
    return poses


def overlapping_visible_view_indexes_per_point(view_indexes_per_point, visible_interval):
    src = np.copy(view_indexes_per_point)
    count = src.shape[1]
    out = np.empty_like(src)
    for i in range(count):
        out[:, i] = np.sum(src[:, max(0, i - visible_interval):min(count, i + visible_interval)], axis=1)
    return out


def quaternion_matrix(quaternion):
    q = np.array(quaternion, dtype=np.float64, copy=True)
    n = np.dot(q, q)
    if n < np.finfo(float).eps * 4.0:
        return np.identity(4)
    q *= np.sqrt(2.0 / n)
    q = np.outer(q, q)
    return np.array([
        [1.0 - q[2, 2] - q[3, 3], q[1, 2] - q[3, 0], q[1, 3] + q[2, 0], 0.0],
        [q[1, 2] + q[3, 0], 1.0 - q[1, 1] - q[3, 3], q[2, 3] - q[1, 0], 0.0],
        [q[1, 3] - q[2, 0], q[2, 3] + q[1, 0], 1.0 - q[1, 1] - q[2, 2], 0.0],
        [0.0, 0.0, 0.0, 1.0]])


def get_extrinsic_matrix_and_projection_matrix(poses, intrinsic_matrix, visible_view_count):
    extrinsics, projections = [], []
    for i in range(visible_view_count):
        pose = poses["poses[%d]" % i]
        o, p = pose["orientation"], pose["position"]
        rigid = quaternion_matrix([o["w"], o["x"], o["y"], o["z"]])
        rigid[0][3], rigid[1][3], rigid[2][3] = p["x"], p["y"], p["z"]
        transform = np.linalg.inv(rigid)
        extrinsics.append(transform)
        projections.append(np.dot(intrinsic_matrix, transform))
    return extrinsics, projections


def global_scale_estimation(extrinsics, point_cloud):
    t = np.stack([np.asarray(e)[:3, 3] for e in extrinsics])
    norm_1 = np.linalg.norm(t.max(0) - t.min(0), ord=2)
    pts = np.asarray([p[:3] for p in point_cloud], dtype=np.float32)
    keep = ~np.any(np.isnan(pts), axis=1)
    keep[0] = True          # utils.py:249-251: the first point initialises the bounds unconditionally
    pts = pts[keep]
    norm_2 = np.linalg.norm(pts.max(0) - pts.min(0), ord=2)
    return max(1.0, max(norm_1, norm_2))


# ---------------------------------------------------------------------------------------------
# cv2 pieces
# ---------------------------------------------------------------------------------------------
def read_bmp_gray(path):
    """cv2.imread(path, cv2.IMREAD_GRAYSCALE) for uncompressed 8-bit paletted and 24-bit BMP files."""
    with open(str(path), "rb") as fp:
        raw = fp.read()
    assert raw[:2] == b"BM"
    data_off = struct.unpack_from("<I", raw, 10)[0]
    hdr_size, width, height, planes, bpp, compression = struct.unpack_from("<IiiHHI", raw, 14)
    assert compression == 0 and bpp in (8, 24)
    colours = struct.unpack_from("<I", raw, 46)[0] or 256
    flip = height > 0
    height = abs(height)
    stride = ((width * bpp + 31) // 32) * 4
    rows = np.frombuffer(raw, np.uint8, stride * height, data_off).reshape(height, stride)
    if bpp == 8:
        pal = np.frombuffer(raw, np.uint8, colours * 4, 14 + hdr_size).reshape(colours, 4).astype(np.int64)      # B G R 0
        gray_of = ((pal[:, 0] * 1868 + pal[:, 1] * 9617 + pal[:, 2] * 4899 + 8192) >> 14).astype(np.uint8)
        img = gray_of[rows[:, :width]]
    else:
        px = rows[:, :3 * width].reshape(height, width, 3).astype(np.int64)
        img = ((px[..., 0] * 1868 + px[..., 1] * 9617 + px[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)
    return img[::-1].copy() if flip else img.copy()


def _linear_taps(dst_size, src_size, scale):
    """(left index, right index, coefficient of left, of right) per destination position -- resize.cpp, INTER_LINEAR."""
    d = np.arange(dst_size)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s).astype(np.float32)
    low = s < 0
    f[low], s[low] = 0.0, 0
    high = s >= src_size - 1
    f[high] = 0.0
    s[high] = src_size - 1
    s1 = np.minimum(s + 1, src_size - 1)
    c0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    return s, s1, c0, c1


def resize_linear(img, downsampling_factor):
    """cv2.resize(img, (0, 0), fx=1/d, fy=1/d) for uint8 images (H, W) or (H, W, C)."""
    inv = 1.0 / downsampling_factor
    h, w = img.shape[:2]
    dh, dw = int(np.rint(h * inv)), int(np.rint(w * inv))          # saturate_cast<int>(ssize * inv_scale)
    scale = 1.0 / inv
    x0, x1, a0, a1 = _linear_taps(dw, w, scale)
    y0, y1, b0, b1 = _linear_taps(dh, h, scale)
    src = img.astype(np.int64)
    shape = (1, dw) + (1,) * (img.ndim - 2)
    rows = src[:, x0] * a0.reshape(shape) + src[:, x1] * a1.reshape(shape)          # horizontal pass, 11 fractional bits
    shape = (dh,) + (1,) * (img.ndim - 1)
    top, bottom = rows[y0] >> 4, rows[y1] >> 4
    out = (((b0.reshape(shape) * top) >> 16) + ((b1.reshape(shape) * bottom) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def erode(img, size=5):
    """cv2.erode(img, np.ones((size, size)), iterations=1): window minimum, pixels outside the image do not count."""
    r = size // 2
    h, w = img.shape
    padded = np.full((h + 2 * r, w + 2 * r), 255, dtype=img.dtype)
    padded[r:r + h, r:r + w] = img
    out = np.full_like(img, 255)
    for dy in range(size):
        for dx in range(size):
            out = np.minimum(out, padded[dy:dy + h, dx:dx + w])
    return out


def downsample_and_crop_mask(mask, downsampling_factor, divide, suggested_h=None, suggested_w=None):
    small = resize_linear(mask, downsampling_factor)
    end_h_index, end_w_index = small.shape
    rows, cols = np.where(small == 255)
    h = rows.max() - rows.min()
    w = cols.max() - cols.min()
    increment_h = divide - h % divide
    increment_w = divide - w % divide
    target_h, target_w = h + increment_h, w + increment_w
    start_h = max(rows.min() - increment_h // 2, 0)
    end_h = start_h + target_h
    start_w = max(cols.min() - increment_w // 2, 0)
    end_w = start_w + target_w
    if suggested_h is not None and suggested_h != h:
        remain = suggested_h - target_h
        start_h = max(start_h - remain // 2, 0)
        end_h = min(suggested_h + start_h, end_h_index)
        start_h = end_h - suggested_h
    if suggested_w is not None and suggested_w != w:
        remain = suggested_w - target_w
        start_w = max(start_w - remain // 2, 0)
        end_w = min(suggested_w + start_w, end_w_index)
        start_w = end_w - suggested_w
    eroded = erode(small, 5)
    return eroded[start_h:end_h, start_w:end_w], int(start_h), int(end_h), int(start_w), int(end_w)


# ---------------------------------------------------------------------------------------------
# JPEG: the library, and the numpy restatement of what follows entropy decoding
# ---------------------------------------------------------------------------------------------
def decode_jpeg_pil(path_or_bytes):
    """RGB uint8 (H, W, 3) from libjpeg-turbo with its defaults (cv2.imread's decoder; channel order aside)."""
    import io
    from PIL import Image
    src = io.BytesIO(path_or_bytes) if isinstance(path_or_bytes, (bytes, bytearray)) else str(path_or_bytes)
    with Image.open(src) as im:
        return np.asarray(im.convert("RGB"))


_FIX = dict(c0_298631336=2446, c0_390180644=3196, c0_541196100=4433, c0_765366865=6270, c0_899976223=7373,
            c1_175875602=9633, c1_501321110=12299, c1_847759065=15137, c1_961570560=16069, c2_053119869=16819,
            c2_562915447=20995, c3_072711026=25172)


def _idct_1d(v, shift):
    """jidctint.c jpeg_idct_islow, one pass over the leading axis of v (8, ...) int64; returns 8 outputs descaled by `shift`."""
    F = _FIX
    z2, z3 = v[2], v[6]
    z1 = (z2 + z3) * F["c0_541196100"]
    tmp2 = z1 + z3 * (-F["c1_847759065"])
    tmp3 = z1 + z2 * F["c0_765366865"]
    z2, z3 = v[0], v[4]
    tmp0 = (z2 + z3) << 13
    tmp1 = (z2 - z3) << 13
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[7], v[5], v[3], v[1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F["c1_175875602"]
    tmp0 = tmp0 * F["c0_298631336"]
    tmp1 = tmp1 * F["c2_053119869"]
    tmp2 = tmp2 * F["c3_072711026"]
    tmp3 = tmp3 * F["c1_501321110"]
    z1 = z1 * (-F["c0_899976223"])
    z2 = z2 * (-F["c2_562915447"])
    z3 = z3 * (-F["c1_961570560"]) + z5
    z4 = z4 * (-F["c0_390180644"]) + z5
    tmp0 += z1 + z3
    tmp1 += z2 + z4
    tmp2 += z2 + z3
    tmp3 += z1 + z4
    rnd = 1 << (shift - 1)
    return [(tmp10 + tmp3 + rnd) >> shift, (tmp11 + tmp2 + rnd) >> shift, (tmp12 + tmp1 + rnd) >> shift, (tmp13 + tmp0 + rnd) >> shift,
            (tmp13 - tmp0 + rnd) >> shift, (tmp12 - tmp1 + rnd) >> shift, (tmp11 - tmp2 + rnd) >> shift, (tmp10 - tmp3 + rnd) >> shift]


def _range_limit(x):
    """idct range-limit table of jdmaster.c prepare_range_limit_table, indexed with (x & 1023), centre offset included."""
    idx = x & 1023
    return np.where(idx < 128, idx + 128, np.where(idx < 512, 255, np.where(idx < 896, 0, idx - 896))).astype(np.uint8)


def idct_islow(blocks, quant):
    """blocks: (..., 8, 8) int coefficients in natural order [row v][column u]; quant (8, 8).  Returns uint8 (..., 8, 8)."""
    c = blocks.astype(np.int64) * quant.astype(np.int64)
    cols = _idct_1d(np.moveaxis(c, -2, 0), 13 - 2)                # pass 1: columns, results scaled up by 2^PASS1_BITS
    ws = np.stack(cols, axis=-2)                                   # (..., 8 rows, 8 columns)
    rows = _idct_1d(np.moveaxis(ws, -1, 0), 13 + 2 + 3)            # pass 2: rows
    return _range_limit(np.stack(rows, axis=-1))


def h2v2_fancy_upsample(plane):
    """jdsample.c h2v2_fancy_upsample on the REAL rows / columns of a chroma plane (h, w) -> (2h, 2w)."""
    p = plane.astype(np.int64)
    h, w = p.shape
    above = np.concatenate([p[:1], p[:-1]])
    below = np.concatenate([p[1:], p[-1:]])
    out = np.empty((2 * h, 2 * w), np.uint8)
    for v, other in ((0, above), (1, below)):
        colsum = 3 * p + other                                     # (h, w)
        last = np.concatenate([colsum[:, :1], colsum[:, :-1]], axis=1)
        nxt = np.concatenate([colsum[:, 1:], colsum[:, -1:]], axis=1)
        even = (colsum * 3 + last + 8) >> 4
        odd = (colsum * 3 + nxt + 7) >> 4
        even[:, 0] = (colsum[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (colsum[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out


def ycc_to_rgb(y, cb, cr):
    """jdcolor.c build_ycc_rgb_table / ycc_rgb_convert."""
    x = np.arange(256, dtype=np.int64) - 128
    fix = lambda v: int(v * 65536 + 0.5)
    cr_r = (fix(1.40200) * x + 32768) >> 16
    cb_b = (fix(1.77200) * x + 32768) >> 16
    cr_g = -fix(0.71414) * x
    cb_g = -fix(0.34414) * x + 32768
    yy = y.astype(np.int64)
    r = np.clip(yy + cr_r[cr], 0, 255)
    g = np.clip(yy + ((cb_g[cb] + cr_g[cr]) >> 16), 0, 255)
    b = np.clip(yy + cb_b[cb], 0, 255)
    return np.stack([r, g, b], axis=-1).astype(np.uint8)


def decode_jpeg_blocks(coefficients, quant, width, height):
    """Everything after entropy decoding for a 3-component 4:2:0 baseline image.
    coefficients: [Y (by, bx, 8, 8), Cb, Cr] block arrays covering the MCU-padded planes; quant: three (8, 8) tables."""
    planes = []
    for blocks, q in zip(coefficients, quant):
        px = idct_islow(blocks, q)                                 # (by, bx, 8, 8)
        planes.append(px.transpose(0, 2, 1, 3).reshape(px.shape[0] * 8, px.shape[1] * 8))
    ch, cw = (height + 1) // 2, (width + 1) // 2
    y = planes[0][:height, :width]
    cb = h2v2_fancy_upsample(planes[1][:ch, :cw])[:height, :width]
    cr = h2v2_fancy_upsample(planes[2][:ch, :cw])[:height, :width]
    return ycc_to_rgb(y, cb, cr)


def bgr_to_hsv_full(img_u8, blue_index=0):
    """cv2.cvtColor(img, cv2.COLOR_BGR2HSV_FULL) (blue_index 0) / COLOR_RGB2HSV_FULL (blue_index 2) on uint8, restated from OpenCV's
    scalar 8-bit path (imgproc/src/color_hsv: RGB2HSV_b with hrange = 256): v = max, s = (diff * sdiv[v] + 2048) >> 12 with
    sdiv[i] = round((255 << 12) / i), h = (hterm * hdiv[diff] + 2048) >> 12 with hdiv[i] = round((256 << 12) / (6 i)) and
    hterm = g - b | b - r + 2 diff | r - g + 4 diff by which channel holds the maximum (r first, then g), + 256 when negative,
    saturated to 8 bits.  PARITY UNPINNED: no cv2-converted image exists in the reference tree (reference utils.py:449-450,
    dataset.py:434-442); known answers below come from the formula (pure red / green / blue -> h = 0 / 85 / 171, s = v = 255)."""
    img = np.asarray(img_u8, dtype=np.uint8)
    b = img[..., blue_index].astype(np.int64)
    g = img[..., 1].astype(np.int64)
    r = img[..., 2 - blue_index].astype(np.int64)
    idx = np.arange(1, 256, dtype=np.float64)
    sdiv = np.zeros(256, dtype=np.int64)
    hdiv = np.zeros(256, dtype=np.int64)
    sdiv[1:] = np.rint((255 << 12) / idx).astype(np.int64)          # saturate_cast<int>(double) = cvRound: half to even
    hdiv[1:] = np.rint((256 << 12) / (6.0 * idx)).astype(np.int64)
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    s = (diff * sdiv[v] + (1 << 11)) >> 12
    hterm = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (hterm * hdiv[diff] + (1 << 11)) >> 12          # arithmetic shift: floor for negatives, as C++ >> on int
    h = np.where(h < 0, h + 256, h)
    return np.stack([np.clip(h, 0, 255), s, v], axis=-1).astype(np.uint8)


def get_pair_color_imgs(prefix_seq, pair_indexes, start_h, end_h, start_w, end_w, downsampling_factor, is_hsv=False, rgb_mode="rgb"):
    """utils.py:441-457: uint8 (2, H, W, 3); is_hsv: the BGR frame through cv2.COLOR_BGR2HSV_FULL (rgb_mode is then not looked at)."""
    imgs = []
    for i in pair_indexes:
        rgb = decode_jpeg_pil(os.path.join(str(prefix_seq), "%08d.jpg" % i))
        small = resize_linear(rgb, downsampling_factor)[start_h:end_h, start_w:end_w, :]
        if is_hsv:
            imgs.append(bgr_to_hsv_full(small[..., ::-1], 0))
        else:
            imgs.append(small if rgb_mode == "rgb" else small[..., ::-1])
    return np.asarray(imgs, dtype=np.uint8)


# ---------------------------------------------------------------------------------------------
# contaminated-point filter -- utils.py:303-404 (compute_sanity_threshold, get_clean_point_list), dataset.py:96-111
# ---------------------------------------------------------------------------------------------
def get_color_imgs(prefix_seq, visible_view_indexes, start_h, end_h, start_w, end_w, downsampling_factor, is_hsv=False):
    """utils.py:288-300: BGR float32 (views, H, W, 3) holding the uint8 values."""
    assert not is_hsv
    imgs = [resize_linear(decode_jpeg_pil(os.path.join(str(prefix_seq), "%08d.jpg" % i)), downsampling_factor)[start_h:end_h, start_w:end_w, ::-1]
            for i in visible_view_indexes]
    return np.array(imgs, dtype="float32")


def bilateral_filter(img, d, sigma_color, sigma_space):
    """cv2.bilateralFilter(src=float32 (H, W, 3), d, sigmaColor, sigmaSpace), BORDER_REFLECT_101 (imgproc/bilateral_filter):
    circular window of radius d // 2, spatial weight exp(-r^2 / (2 sigma_space^2)), colour weight exp(-(|db| + |dg| + |dr|)^2 /
    (2 sigma_color^2)) shared by the channels.  OpenCV evaluates the colour weight through a 4096-bin-per-channel table with linear
    interpolation; the closed form differs from it by ~1e-7 relative."""
    radius = d // 2
    src = img.astype(np.float32)
    padded = np.pad(src, ((radius, radius), (radius, radius), (0, 0)), mode="reflect")
    h, w, _ = src.shape
    num = np.zeros_like(src, dtype=np.float64)
    den = np.zeros((h, w, 1), dtype=np.float64)
    cc, sc = -0.5 / (sigma_color * sigma_color), -0.5 / (sigma_space * sigma_space)
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            r = np.sqrt(float(i * i + j * j))
            if r > radius:
                continue
            nb = padded[radius + i:radius + i + h, radius + j:radius + j + w].astype(np.float64)
            l1 = np.abs(nb - src).sum(axis=2, keepdims=True)
            wgt = np.exp(r * r * sc) * np.exp(l1 * l1 * cc)
            num += wgt * nb
            den += wgt
    return (num / den).astype(np.float32)


def compute_sanity_threshold(sanity_array, inlier_percentage):
    """utils.py:303-336: grow a window around the histogram's peak bin until it holds `inlier_percentage` of the samples."""
    hist, edges = np.histogram(sanity_array, bins=np.arange(1000) * np.max(sanity_array) / 1000.0, density=True)
    share = hist * np.diff(edges)
    peak = int(np.argmax(share))
    total = share[peak]
    up, down = 1, 1
    while True:
        if peak + up < len(share):
            total += share[peak + up]
            up += 1
            if total >= inlier_percentage:
                return edges[peak - down + 1], edges[peak + up]
        if peak - down >= 0:
            total += share[peak - down]
            down += 1
            if total >= inlier_percentage:
                return edges[peak - down + 1], edges[peak + up]
        if peak + up >= len(share) and peak - down < 0:
            return np.min(edges), np.max(edges)


def point_brightness(img_bgr_u8values, d=7, sigma_color=25, sigma_space=25):
    """V of cv2.COLOR_BGR2HSV_FULL (= max(B, G, R) for float images) of the bilateral-filtered frame / 255 (utils.py:352-356)."""
    return bilateral_filter(np.asarray(img_bgr_u8values, dtype=np.float32) / 255.0, d, sigma_color, sigma_space).max(axis=2)


def frame_point_terms(img, pts, view_column, mask, projection_matrix, extrinsic_matrix, brightness=None):
    """One frame of utils.py:345-388: (indices of the points that are visible, project inside the image and the mask; their
    camera depth; the filtered brightness at their pixel)."""
    height, width = np.asarray(img).shape[:2]
    value = (brightness if brightness is not None else point_brightness(img)).reshape(-1)
    visible = np.where(np.asarray(view_column).reshape(-1) > 0.5)[0]
    cam = np.einsum('ij,mj->mi', np.asarray(extrinsic_matrix), pts)
    cam = cam / cam[:, 3].reshape((-1, 1))
    px = np.einsum('ij,mj->mi', np.asarray(projection_matrix), pts)
    px = px / px[:, 2].reshape((-1, 1))
    vpx, vcam = px[visible].reshape((-1, 3)), cam[visible].reshape((-1, 4))
    inside = np.where((vpx[:, 0] <= width - 1) & (vpx[:, 0] >= 0) & (vpx[:, 1] <= height - 1) & (vpx[:, 1] >= 0) & (vcam[:, 2] > 0))[0]
    loc = (np.round(vpx[inside, 0]) + np.round(vpx[inside, 1]) * width).astype(np.int32).reshape(-1)
    in_mask = np.where(np.asarray(mask).reshape(-1)[loc] == 255)[0]
    return visible[inside[in_mask]], vcam[inside[in_mask], 2], value[loc[in_mask]]


def get_clean_point_list(imgs, point_cloud, view_indexes_per_point, mask_boundary, inlier_percentage, projection_matrices,
                         extrinsic_matrices, is_hsv=False, brightness=None):
    """utils.py:339-404 for is_hsv False.  brightness: optional per-frame (H, W) planes replacing the filtered V channel."""
    assert not is_hsv
    pts = np.asarray(point_cloud).reshape((-1, 4))
    if inlier_percentage <= 0.0 or inlier_percentage >= 1.0:
        return list()
    contaminated = np.zeros(pts.shape[0], dtype=np.int32)
    appearances = np.zeros(pts.shape[0], dtype=np.int32)
    for i in range(len(projection_matrices)):
        index, depth, value = frame_point_terms(imgs[i], pts, view_indexes_per_point[:, i], mask_boundary, projection_matrices[i],
                                                extrinsic_matrices[i], None if brightness is None else brightness[i])
        sanity = depth ** 2 * value
        appearances[index] += 1
        if sanity.shape[0] < 2:
            continue
        lo, hi = compute_sanity_threshold(sanity, inlier_percentage)
        contaminated[index[(sanity <= lo) | (sanity >= hi)]] += 1
    return (contaminated < appearances / 2).astype(np.float32)
