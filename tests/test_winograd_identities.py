"""The Winograd transforms the HIP kernels hard-code, checked on the CPU in float64 against the sums they stand for (no GPU, no oracle):

  * F(3x3, 4x4), the dense layers' weight gradient (csrc/wgrad_f34_kernels.h): a 6 x 6 activation patch d and a 4 x 4 tile g of the output
    gradient give the tile's contribution to the 3 x 3 taps as A^T [ (S g S^T) .* (B^T d B) ] A, with the scales of S folded into the
    output transform exactly as the kernel does (f34_s has no scales, C = A^T diag(s));
  * F(4x4, 3x3), the level-0 dense-layer forward (csrc/wino4_fwd_kernels.h): Y = A^T [ (G w G^T) .* (B^T d B) ] A with its own interpolation
    points 0, +-5/8, +-3/2, inf (w4_bt / w4_g / w4_at below).

The arithmetic below restates the kernels' helper functions (f34_bt, f34_s, w4_bt, w4_at, the weights kernel's G) line by line, so a typo in a
constant there shows up here as well as in the GPU parity tests."""
import numpy as np


def f34_bt(d):          # 6-point input transform B^T d (wgrad_f34_kernels.h: f34_bt)
    d0, d1, d2, d3, d4, d5 = d
    pp, qq = d4 - 4 * d2, d3 - 4 * d1
    rr, ss = d4 - d2, d3 - d1
    return np.array([4 * d0 - 5 * d2 + d4, pp + qq, pp - qq, rr + 2 * ss, rr - 2 * ss, 4 * d1 - 5 * d3 + d5])


def f34_s(g):           # 4 -> 6 transform of the gradient tile WITHOUT the row scales (f34_s)
    g0, g1, g2, g3 = g
    e, o = g0 + g2, g1 + g3
    e4, o4 = g0 + 4 * g2, g1 + 4 * g3
    return np.array([g0, e + o, e - o, e4 + 2 * o4, e4 - 2 * o4, g3])


# output transform of the weight gradient with the scales (1/4, -1/6, -1/6, 1/24, 1/24, 1) of S folded in (the kernel's C)
C = np.array([[0.25, -1 / 6, -1 / 6, 1 / 24, 1 / 24, 0.0],
              [0.0, -1 / 6, 1 / 6, 1 / 12, -1 / 12, 0.0],
              [0.0, -1 / 6, -1 / 6, 1 / 6, 1 / 6, 1.0]])


# the forward's interpolation points +-a, +-b (wino4_fwd_kernels.h: kW4A, kW4B and the constants derived from them)
W4A, W4B = 0.625, 1.5
W4A2, W4B2 = W4A * W4A, W4B * W4B
W4P, W4S = W4A2 * W4B2, W4A2 + W4B2
W4N0, W4NA, W4NB = 1 / W4P, 1 / (2 * W4A2 * (W4A2 - W4B2)), 1 / (2 * W4B2 * (W4B2 - W4A2))


def w4_bt(d):           # 6-point input transform B^T d of the forward (wino4_fwd_kernels.h: w4_bt)
    d0, d1, d2, d3, d4, d5 = d
    pp, qh = d4 - W4B2 * d2, d3 - W4B2 * d1
    rr, sh = d4 - W4A2 * d2, d3 - W4A2 * d1
    return np.array([W4P * d0 - W4S * d2 + d4, pp + W4A * qh, pp - W4A * qh, rr + W4B * sh, rr - W4B * sh, W4P * d1 - W4S * d3 + d5])


def w4_at(m):           # 6 -> 4 output transform A^T m of the forward (wino4_fwd_kernels.h: w4_at)
    m0, m1, m2, m3, m4, m5 = m
    s1, d1, s2, d2 = m1 + m2, m1 - m2, m3 + m4, m3 - m4
    return np.array([m0 + s1 + s2, W4A * d1 + W4B * d2, W4A2 * s1 + W4B2 * s2, W4A2 * W4A * d1 + W4B2 * W4B * d2 + m5])


def w4_g(w):            # 3 -> 6 filter transform G w (wino4_fwd_weights_kernel)
    a, b, c = w
    ea, eb = a + W4A2 * c, a + W4B2 * c
    return np.array([W4N0 * a, W4NA * (ea + W4A * b), W4NA * (ea - W4A * b), W4NB * (eb + W4B * b), W4NB * (eb - W4B * b), c])


def two_d(f, m, n_out):
    """apply the 1-D transform f along both axes of m"""
    cols = np.stack([f(m[:, c]) for c in range(m.shape[1])], axis=1)
    return np.stack([f(cols[i, :]) for i in range(n_out)], axis=0)


def test_weight_gradient_f34_identity():
    rng = np.random.default_rng(3)
    for _ in range(20):
        d = rng.standard_normal((6, 6))          # activation patch: rows / columns -1 .. 4 around the tile
        g = rng.standard_normal((4, 4))          # output-gradient tile
        want = np.array([[sum(d[ky + y, kx + x] * g[y, x] for y in range(4) for x in range(4)) for kx in range(3)] for ky in range(3)])
        v = two_d(f34_bt, d, 6)
        u = two_d(f34_s, g, 6)
        got = C @ (u * v) @ C.T
        assert np.abs(got - want).max() < 1e-12 * max(1.0, np.abs(want).max())


def test_weight_gradient_f34_sums_before_the_output_transform():
    """the kernel adds the products of MANY tiles per transform-domain position and transforms the sum once (linearity)"""
    rng = np.random.default_rng(4)
    acc = np.zeros((6, 6))
    want = np.zeros((3, 3))
    for _ in range(50):
        d = rng.standard_normal((6, 6))
        g = rng.standard_normal((4, 4))
        want += np.array([[sum(d[ky + y, kx + x] * g[y, x] for y in range(4) for x in range(4)) for kx in range(3)] for ky in range(3)])
        acc += two_d(f34_s, g, 6) * two_d(f34_bt, d, 6)
    assert np.abs(C @ acc @ C.T - want).max() < 1e-11 * np.abs(want).max()


def test_forward_f43_identity():
    rng = np.random.default_rng(5)
    for _ in range(20):
        d = rng.standard_normal((6, 6))
        w = rng.standard_normal((3, 3))
        want = np.array([[sum(d[y + ky, x + kx] * w[ky, kx] for ky in range(3) for kx in range(3)) for x in range(4)] for y in range(4)])
        got = two_d(w4_at, two_d(w4_g, w, 6) * two_d(w4_bt, d, 6), 4)
        assert np.abs(got - want).max() < 1e-12 * max(1.0, np.abs(want).max())


def test_forward_f43_points_round_better_than_the_textbook_ones():
    """Why the forward does not use the points 0, +-1, +-2, inf of the weight gradient: one layer's worth of fp32 arithmetic (sequential fp32 sum
    over 180 channels of transform-domain products, as the MFMA K loop forms it) against fp64, both point sets on the same data."""
    rng = np.random.default_rng(8)
    k, tiles = 180, 64
    x = np.maximum(rng.standard_normal((tiles, k, 6, 6)), 0.0)
    w = rng.standard_normal((k, 3, 3)) * (2.0 / (k * 9)) ** 0.5
    want = np.stack([np.einsum("tkab,kab->t", x[:, :, i:i + 3, j:j + 3], w) for i in range(4) for j in range(4)], axis=1).reshape(tiles, 4, 4)

    def textbook_g(v):
        a, b, c = v
        return np.array([0.25 * a, (-1 / 6) * (a + b + c), (-1 / 6) * (a - b + c), (1 / 24) * (a + 2 * b + 4 * c), (1 / 24) * (a - 2 * b + 4 * c), c])

    def textbook_at(m):
        m0, m1, m2, m3, m4, m5 = m
        s1, d1, s2, d2 = m1 + m2, m1 - m2, m3 + m4, m3 - m4
        return np.array([m0 + s1 + s2, d1 + 2 * d2, s1 + 4 * s2, d1 + 8 * d2 + m5])

    errs = []
    for bt_f, g_f, at_f in ((w4_bt, w4_g, w4_at), (f34_bt, textbook_g, textbook_at)):
        f32 = np.float32
        bt = np.array([bt_f(row) for row in np.eye(6)]).T.astype(f32)
        gm = np.array([g_f(row) for row in np.eye(3)]).T.astype(f32)
        at = np.array([at_f(row) for row in np.eye(6)]).T.astype(f32)
        u = np.einsum("ia,kab,jb->kij", gm, w.astype(f32), gm).astype(f32)
        v = np.einsum("ia,tkab,jb->tkij", bt, x.astype(f32), bt).astype(f32)
        m = np.zeros((tiles, 6, 6), f32)
        for c in range(k):
            m = (m + u[c][None] * v[:, c]).astype(f32)
        y = np.einsum("ia,tab,jb->tij", at, m, at).astype(f32)
        errs.append(float(np.sqrt(np.mean((y - want) ** 2)) / np.abs(want).max()))
    assert errs[0] < 0.7 * errs[1], errs          # measured 0.45 x


def test_fp32_rounding_of_the_weight_gradient_form():
    """fp32 evaluation of the F(3x3, 4x4) sums over a 256 x 320 plane against fp64: the direct sum's error level, not the 1e-4 .. 1e-3 that
    larger-tile Winograd forms are known for in the FORWARD direction (here the long sum over tiles dominates either way)."""
    rng = np.random.default_rng(6)
    h, w = 128, 160
    x = np.maximum(rng.standard_normal((h + 2, w + 2)) + 0.3, 0.0)
    x[0, :] = x[-1, :] = 0.0
    x[:, 0] = x[:, -1] = 0.0
    g = rng.standard_normal((h, w)) * 1e-3
    want = np.array([[(x[ky:ky + h, kx:kx + w] * g).sum() for kx in range(3)] for ky in range(3)])
    acc = np.zeros((6, 6), np.float32)
    x32, g32 = x.astype(np.float32), g.astype(np.float32)
    bt = np.array([f34_bt(row) for row in np.eye(6)]).T.astype(np.float32)          # matrices of the two transforms
    sm = np.array([f34_s(row) for row in np.eye(4)]).T.astype(np.float32)
    th, tw = h // 4, w // 4
    patches = np.lib.stride_tricks.sliding_window_view(x32, (6, 6))[::4, ::4][:th, :tw]
    tiles = g32.reshape(th, 4, tw, 4).transpose(0, 2, 1, 3)
    v = np.einsum("ij,abjk,lk->abil", bt, patches, bt).astype(np.float32)
    u = np.einsum("ij,abjk,lk->abil", sm, tiles, sm).astype(np.float32)
    prod = (u * v).astype(np.float32)
    for a in range(th):          # fp32 running sums, one per tile row, then a short fp32 tree -- like a wave's accumulators and the reduce
        acc = (acc + prod[a].sum(axis=0, dtype=np.float32)).astype(np.float32)
    got = C @ acc.astype(np.float64) @ C.T
    assert np.abs(got - want).max() < 2e-5 * np.abs(want).max()
