// Differentiable geometry layers: depth scaling, flow-from-depth, depth warping.
// HBM-bound streaming / gather kernels: every plane is read once per pass, the
// per-sample camera algebra is recomputed per block in fp64 (27 flops) instead of being
// materialised, reductions use wave shuffles + one fp64 atomic per block.
#include "common.h"

namespace endo {

// ------------------------------------------------------------------------------------------
// camera maps (reference models.py:391-399 / 492-499 / 531-532)
//   M  = K R^T K^-1,  w  = -K R^T t          (frame-1 pixel + depth -> frame-2 homogeneous pixel)
//   M2 = K R   K^-1,  w2 =  K t              (only the z row / z entry is ever used)
// ------------------------------------------------------------------------------------------
struct Camera {
    float m[9];
    float w[3];
    float m2z[3];
    float w2z;
};

__device__ inline void mat3_mul(const double* a, const double* b, double* c) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

__device__ inline void camera_setup(const float* K, const float* R, const float* t, Camera* cam) {
    double k[9], r[9], rt[9], ki[9], tv[3];
    for (int i = 0; i < 9; ++i) { k[i] = K[i]; r[i] = R[i]; }
    for (int i = 0; i < 3; ++i) { tv[i] = t[i]; for (int j = 0; j < 3; ++j) rt[i * 3 + j] = r[j * 3 + i]; }
    const double det = k[0] * (k[4] * k[8] - k[5] * k[7]) - k[1] * (k[3] * k[8] - k[5] * k[6]) +
                       k[2] * (k[3] * k[7] - k[4] * k[6]);
    const double id = 1.0 / det;
    ki[0] = (k[4] * k[8] - k[5] * k[7]) * id; ki[1] = (k[2] * k[7] - k[1] * k[8]) * id; ki[2] = (k[1] * k[5] - k[2] * k[4]) * id;
    ki[3] = (k[5] * k[6] - k[3] * k[8]) * id; ki[4] = (k[0] * k[8] - k[2] * k[6]) * id; ki[5] = (k[2] * k[3] - k[0] * k[5]) * id;
    ki[6] = (k[3] * k[7] - k[4] * k[6]) * id; ki[7] = (k[1] * k[6] - k[0] * k[7]) * id; ki[8] = (k[0] * k[4] - k[1] * k[3]) * id;
    double krt[9], m[9], kr[9], m2[9];
    mat3_mul(k, rt, krt);
    mat3_mul(krt, ki, m);
    mat3_mul(k, r, kr);
    mat3_mul(kr, ki, m2);
    for (int i = 0; i < 9; ++i) cam->m[i] = static_cast<float>(m[i]);
    for (int i = 0; i < 3; ++i)
        cam->w[i] = static_cast<float>(-(krt[i * 3] * tv[0] + krt[i * 3 + 1] * tv[1] + krt[i * 3 + 2] * tv[2]));
    for (int j = 0; j < 3; ++j) cam->m2z[j] = static_cast<float>(m2[6 + j]);
    cam->w2z = static_cast<float>(k[6] * tv[0] + k[7] * tv[1] + k[8] * tv[2]);
}

__device__ __forceinline__ void load_camera(const float* K, const float* R, const float* t, int n, Camera* shared_cam) {
    if (threadIdx.x == 0) camera_setup(K + 9 * n, R + 9 * n, t + 3 * n, shared_cam);
    __syncthreads();
}

__device__ __forceinline__ void ray(const Camera& c, float x, float y, float& qx, float& qy, float& qz) {
    qx = fmaf(c.m[1], y, c.m[0] * x) + c.m[2];
    qy = fmaf(c.m[4], y, c.m[3] * x) + c.m[5];
    qz = fmaf(c.m[7], y, c.m[6] * x) + c.m[8];
}

// ------------------------------------------------------------------------------------------
// depth scaling (models.py:346-363)
// ------------------------------------------------------------------------------------------
constexpr int kRedThreads = 256;
constexpr int kRedItems = 8;   // pixels per thread per block

__global__ void __launch_bounds__(kRedThreads) depth_scale_pass1(const float* __restrict__ sd,
                                                                  const float* __restrict__ sm,
                                                                  double* stats, int hw) {
    __shared__ double scratch[2 * (kRedThreads / 64)];
    const int n = blockIdx.y;
    const int64_t base = static_cast<int64_t>(n) * hw;
    float part[2] = {0.f, 0.f};
    for (int i = blockIdx.x * kRedThreads * kRedItems + threadIdx.x, k = 0; k < kRedItems && i < hw; ++k, i += kRedThreads) {
        const float b = sm[base + i] > 1.0e-8f ? 1.f : 0.f;
        part[0] += sd[base + i] * b;
        part[1] += b;
    }
    block_sum_atomic<2>(part, stats + 8 * n, scratch);
}

__global__ void __launch_bounds__(kRedThreads) depth_scale_pass2(const float* __restrict__ pred,
                                                                  const float* __restrict__ sd,
                                                                  double* stats, int hw, float eps) {
    __shared__ double scratch[3 * (kRedThreads / 64)];
    const int n = blockIdx.y;
    const int64_t base = static_cast<int64_t>(n) * hw;
    const float mean_sd = static_cast<float>(stats[8 * n + 0]) / static_cast<float>(stats[8 * n + 1]);
    const float thr = 0.5f * mean_sd;
    float part[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * kRedThreads * kRedItems + threadIdx.x, k = 0; k < kRedItems && i < hw; ++k, i += kRedThreads) {
        const float s = sd[base + i];
        if (s > thr) {
            const float v = s / (eps + pred[base + i]);
            part[0] += v;
            part[1] += 1.f;
            part[2] += v * v;
        }
    }
    block_sum_atomic<3>(part, stats + 8 * n + 2, scratch);
}

// per-sample scale / std from the sums; also the (N x N)-broadcast ratio of models.py:363
__global__ void depth_scale_finalize(double* stats, float* ratio, int n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double inv_mean = 0.0, std_mean = 0.0;
    for (int i = 0; i < n; ++i) {
        double* s = stats + 8 * i;
        const double a = s[3];
        const double scale = s[2] / a;
        double var = (s[4] - scale * scale * a) / a;
        if (var < 0.0) var = 0.0;
        const double sd = sqrt(var);
        s[5] = scale;
        s[6] = sd;
        inv_mean += 1.0 / scale;
        std_mean += sd;
    }
    inv_mean /= n;
    std_mean /= n;
    stats[7] = inv_mean;            // mean_i 1/scale_i
    if (n > 1) stats[8 + 7] = std_mean;   // mean_j std_j   (kept in sample 1's spare slot)
    *ratio = static_cast<float>(inv_mean * std_mean);
}

__global__ void __launch_bounds__(256) depth_scale_apply(const float* __restrict__ pred, const double* __restrict__ stats,
                                                         float* __restrict__ scaled, int hw) {
    const int n = blockIdx.y;
    const float scale = static_cast<float>(stats[8 * n + 5]);
    const int64_t base = static_cast<int64_t>(n) * hw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x)
        scaled[base + i] = scale * pred[base + i];
}

__global__ void __launch_bounds__(kRedThreads) depth_scale_bwd_reduce(const float* __restrict__ g,
                                                                      const float* __restrict__ pred,
                                                                      double* work, int hw) {
    __shared__ double scratch[kRedThreads / 64];
    const int n = blockIdx.y;
    const int64_t base = static_cast<int64_t>(n) * hw;
    float part[1] = {0.f};
    for (int i = blockIdx.x * kRedThreads * kRedItems + threadIdx.x, k = 0; k < kRedItems && i < hw; ++k, i += kRedThreads)
        part[0] += g[base + i] * pred[base + i];
    block_sum_atomic<1>(part, work + n, scratch);
}

__global__ void __launch_bounds__(256) depth_scale_bwd_apply(const float* __restrict__ g, const float* __restrict__ grad_ratio,
                                                             const float* __restrict__ pred, const float* __restrict__ sd,
                                                             const double* __restrict__ stats, const double* __restrict__ work,
                                                             float* __restrict__ grad_pred, int nsamples, int hw, float eps) {
    const int n = blockIdx.y;
    const double* s = stats + 8 * n;
    const float mean_sd = static_cast<float>(s[0]) / static_cast<float>(s[1]);
    const float thr = 0.5f * mean_sd;
    const double a = s[3];
    const double scale = s[5];
    const double sdev = s[6];
    // dL/dsmap_p = c0 + c1 * (smap_p - scale)
    double c0 = (g ? work[n] : 0.0) / a;
    double c1 = 0.0;
    if (grad_ratio) {
        const double gr = static_cast<double>(*grad_ratio);
        const double inv_mean = stats[7];
        const double std_mean = nsamples > 1 ? stats[8 + 7] : sdev;
        c1 = gr * inv_mean / (nsamples * sdev * a);
        c0 -= gr * std_mean / (nsamples * scale * scale * a);
    }
    const float fscale = static_cast<float>(scale);
    const int64_t base = static_cast<int64_t>(n) * hw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const float p = pred[base + i];
        float out = g ? fscale * g[base + i] : 0.f;
        const float sv = sd[base + i];
        if (sv > thr) {
            const float den = eps + p;
            const float smap = sv / den;
            const float coef = static_cast<float>(c0 + c1 * (static_cast<double>(smap) - scale));
            out += coef * (-smap / den);
        }
        grad_pred[base + i] = out;
    }
}

// ------------------------------------------------------------------------------------------
// flow from depth (models.py:377-451)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) flow_fwd_kernel(const float* __restrict__ depth, const float* __restrict__ mask,
                                                       const float* __restrict__ t, const float* __restrict__ R,
                                                       const float* __restrict__ K, float* __restrict__ flow,
                                                       int h, int w) {
    __shared__ Camera cam;
    const int n = blockIdx.y;
    load_camera(K, R, t, n, &cam);
    const int hw = h * w;
    const int64_t base = static_cast<int64_t>(n) * hw;
    const float fw = static_cast<float>(w), fh = static_cast<float>(h);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const int yy = i / w, xx = i - yy * w;
        const float x = static_cast<float>(xx), y = static_cast<float>(yy);
        float qx, qy, qz;
        ray(cam, x, y, qx, qy, qz);
        const float d = depth[base + i], m = mask[base + i];
        const float z2 = cam.w[2] + d * qz;
        const float zt = 1.0e30f * (1.0f - m) + m * z2;
        const float u2 = (cam.w[0] + d * qx) / zt;
        const float v2 = (cam.w[1] + d * qy) / zt;
        flow[2 * base + i] = (u2 - x) / fw;
        flow[2 * base + hw + i] = (v2 - y) / fh;
    }
}

__global__ void __launch_bounds__(256) flow_bwd_kernel(const float* __restrict__ gflow, const float* __restrict__ depth,
                                                       const float* __restrict__ mask, const float* __restrict__ t,
                                                       const float* __restrict__ R, const float* __restrict__ K,
                                                       float* __restrict__ gdepth, int h, int w) {
    __shared__ Camera cam;
    const int n = blockIdx.y;
    load_camera(K, R, t, n, &cam);
    const int hw = h * w;
    const int64_t base = static_cast<int64_t>(n) * hw;
    const float fw = static_cast<float>(w), fh = static_cast<float>(h);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const int yy = i / w, xx = i - yy * w;
        float qx, qy, qz;
        ray(cam, static_cast<float>(xx), static_cast<float>(yy), qx, qy, qz);
        const float d = depth[base + i], m = mask[base + i];
        const float z2 = cam.w[2] + d * qz;
        const float zt = 1.0e30f * (1.0f - m) + m * z2;
        const float nx = cam.w[0] + d * qx;
        const float ny = cam.w[1] + d * qy;
        const float gu = gflow[2 * base + i] / fw;
        const float gv = gflow[2 * base + hw + i] / fh;
        // u2 = nx / zt :  d u2 / d d = qx / zt - nx / zt^2 * (m qz)
        const float gzt = -(gu * nx + gv * ny) / (zt * zt);
        gdepth[base + i] = gu * qx / zt + gv * qy / zt + gzt * m * qz;
    }
}

// ------------------------------------------------------------------------------------------
// depth warping (models.py:469-554) with the grid_sample of models.py:325-336 folded in.
// Source location of the CPU grid_sample path: ix = (gx + 1) * (W / 2) - 0.5, gx = 2 (u / W) - 1.
// ------------------------------------------------------------------------------------------
struct Taps {
    float wnw, wne, wsw, wse;   // bilinear weights
    int x0, y0;                 // north-west tap (valid flags say which taps are in range)
    bool vw, ve, vn, vs;        // column west/east, row north/south in range
    float fx, fy;               // fractional parts (w, n in ATen's naming)
};

__device__ __forceinline__ Taps make_taps(float u2, float v2, int w, int h) {
    Taps tp;
    const float fw = static_cast<float>(w), fh = static_cast<float>(h);
    const float gx = 2.0f * (u2 / fw) - 1.0f;
    const float gy = 2.0f * (v2 / fh) - 1.0f;
    const float ix = (gx + 1.0f) * (fw * 0.5f) - 0.5f;
    const float iy = (gy + 1.0f) * (fh * 0.5f) - 0.5f;
    const float xw = floorf(ix), yn = floorf(iy);
    const float wx = ix - xw, ee = 1.0f - wx;
    const float ny = iy - yn, ss = 1.0f - ny;
    tp.wnw = ss * ee; tp.wne = ss * wx; tp.wsw = ny * ee; tp.wse = ny * wx;
    tp.fx = wx; tp.fy = ny;
    // range tests in float so that huge / non-finite coordinates never reach an int conversion
    tp.vw = (xw >= 0.0f) && (xw <= fw - 1.0f);
    tp.ve = (xw + 1.0f >= 0.0f) && (xw + 1.0f <= fw - 1.0f);
    tp.vn = (yn >= 0.0f) && (yn <= fh - 1.0f);
    tp.vs = (yn + 1.0f >= 0.0f) && (yn + 1.0f <= fh - 1.0f);
    const bool any = (tp.vw || tp.ve) && (tp.vn || tp.vs);
    tp.x0 = any ? static_cast<int>(xw) : 0;
    tp.y0 = any ? static_cast<int>(yn) : 0;
    return tp;
}

// (M2 p)_z at frame-2 pixel (xx, yy)
__device__ __forceinline__ float plane_s(const Camera& c, int xx, int yy) {
#pragma clang fp contract(off)
    return fmaf(c.m2z[1], static_cast<float>(yy), c.m2z[0] * static_cast<float>(xx)) + c.m2z[2];
}

// depth 2 seen from camera 1 at frame-2 pixel (xx, yy):  m * (w2z + (d2 m) * (M2 p)_z)
__device__ __forceinline__ float depth_in_1(const Camera& c, const float* d2, const float* mask, int64_t base, int w, int xx, int yy,
                                            float* mask_out, float* s_out) {
#pragma clang fp contract(off)
    const int64_t o = base + static_cast<int64_t>(yy) * w + xx;
    const float m = mask[o];
    const float s = plane_s(c, xx, yy);
    *mask_out = m;
    *s_out = s;
    return m * fmaf(d2[o] * m, s, c.w2z);
}

// The tap blend and the d1 gradient, shared by the gather and the LDS-staged kernels so that a tile shape changes speed and
// nothing else: explicit fused multiply-adds in a fixed order (contraction off), identical in every kernel that inlines them.
__device__ __forceinline__ void tap_blend(const Taps& q, const float (&v)[4], const float (&mm)[4], float& acc, float& macc) {
#pragma clang fp contract(off)
    acc = q.wnw * v[0]; macc = q.wnw * mm[0];
    acc = fmaf(q.wne, v[1], acc); macc = fmaf(q.wne, mm[1], macc);
    acc = fmaf(q.wsw, v[2], acc); macc = fmaf(q.wsw, mm[2], macc);
    acc = fmaf(q.wse, v[3], acc); macc = fmaf(q.wse, mm[3], macc);
}

__device__ __forceinline__ float warp_grad_d1(const Taps& q, const float (&v)[4], float g, float fw, float fh, float qx, float qy, float qz,
                                              float zt, float nx, float ny, bool open, float m) {
#pragma clang fp contract(off)
    const float sfrac = 1.0f - q.fy, efrac = 1.0f - q.fx;
    // d out / d ix, d out / d iy, then grid_sample's (W/2, H/2) and the grid's (2/W, 2/H)
    const float gix = fmaf(v[3] - v[2], q.fy, (v[1] - v[0]) * sfrac) * g;
    const float giy = fmaf(v[3] - v[1], q.fx, (v[2] - v[0]) * efrac) * g;
    const float gu = gix * (fw * 0.5f) * 2.0f / fw;
    const float gv = giy * (fh * 0.5f) * 2.0f / fh;
    float gdm = fmaf(gv, qy / zt, gu * qx / zt);
    if (open) gdm = fmaf(-fmaf(gv, ny, gu * nx) / (zt * zt), qz, gdm);
    return gdm * m;
}

__device__ __forceinline__ float warp_grad_d2_term(float g, float wt, float mm, float ss) {
#pragma clang fp contract(off)
    return g * wt * mm * ss * mm;
}

__global__ void __launch_bounds__(256) warp_fwd_kernel(const float* __restrict__ d1, const float* __restrict__ d2,
                                                       const float* __restrict__ mask, const float* __restrict__ t,
                                                       const float* __restrict__ R, const float* __restrict__ K,
                                                       float* __restrict__ warped, float* __restrict__ intersect,
                                                       int h, int w, float eps) {
    __shared__ Camera cam;
    const int n = blockIdx.y;
    load_camera(K, R, t, n, &cam);
    const int hw = h * w;
    const int64_t base = static_cast<int64_t>(n) * hw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const int yy = i / w, xx = i - yy * w;
        float qx, qy, qz;
        ray(cam, static_cast<float>(xx), static_cast<float>(yy), qx, qy, qz);
        const float m = mask[base + i];
        const float dm = d1[base + i] * m;
        float zt = cam.w[2] + dm * qz;
        zt = (m > 0.5f) ? zt : eps;
        zt = (zt > 0.0f) ? zt : eps;
        const float u2 = (cam.w[0] + dm * qx) / zt;
        const float v2 = (cam.w[1] + dm * qy) / zt;
        const Taps tp = make_taps(u2, v2, w, h);
        float v[4] = {0.f, 0.f, 0.f, 0.f}, mm[4] = {0.f, 0.f, 0.f, 0.f}, ss;
        if (tp.vw && tp.vn) v[0] = depth_in_1(cam, d2, mask, base, w, tp.x0, tp.y0, &mm[0], &ss);
        if (tp.ve && tp.vn) v[1] = depth_in_1(cam, d2, mask, base, w, tp.x0 + 1, tp.y0, &mm[1], &ss);
        if (tp.vw && tp.vs) v[2] = depth_in_1(cam, d2, mask, base, w, tp.x0, tp.y0 + 1, &mm[2], &ss);
        if (tp.ve && tp.vs) v[3] = depth_in_1(cam, d2, mask, base, w, tp.x0 + 1, tp.y0 + 1, &mm[3], &ss);
        float acc, macc;
        tap_blend(tp, v, mm, acc, macc);
        warped[base + i] = acc;
        intersect[base + i] = (macc * m >= 0.9f) ? 1.0f : 0.0f;
    }
}

__global__ void __launch_bounds__(256) warp_bwd_kernel(const float* __restrict__ gw, const float* __restrict__ d1,
                                                       const float* __restrict__ d2, const float* __restrict__ mask,
                                                       const float* __restrict__ t, const float* __restrict__ R,
                                                       const float* __restrict__ K, float* __restrict__ gd1,
                                                       float* gd2, int h, int w, float eps) {
    __shared__ Camera cam;
    const int n = blockIdx.y;
    load_camera(K, R, t, n, &cam);
    const int hw = h * w;
    const int64_t base = static_cast<int64_t>(n) * hw;
    const float fw = static_cast<float>(w), fh = static_cast<float>(h);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const int yy = i / w, xx = i - yy * w;
        float qx, qy, qz;
        ray(cam, static_cast<float>(xx), static_cast<float>(yy), qx, qy, qz);
        const float m = mask[base + i];
        const float dm = d1[base + i] * m;
        const float z2 = cam.w[2] + dm * qz;
        float zt = (m > 0.5f) ? z2 : eps;
        const bool open = (m > 0.5f) && (zt > 0.0f);
        zt = (zt > 0.0f) ? zt : eps;
        const float nx = cam.w[0] + dm * qx;
        const float ny = cam.w[1] + dm * qy;
        const Taps tp = make_taps(nx / zt, ny / zt, w, h);
        const float g = gw[base + i];
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const bool val[4] = {tp.vw && tp.vn, tp.ve && tp.vn, tp.vw && tp.vs, tp.ve && tp.vs};
        const float wt[4] = {tp.wnw, tp.wne, tp.wsw, tp.wse};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (!val[c]) continue;
            const int sx = tp.x0 + (c & 1), sy = tp.y0 + (c >> 1);
            float mm, ss;
            v[c] = depth_in_1(cam, d2, mask, base, w, sx, sy, &mm, &ss);
            atomicAdd(gd2 + base + static_cast<int64_t>(sy) * w + sx, warp_grad_d2_term(g, wt[c], mm, ss));
        }
        gd1[base + i] = warp_grad_d1(tp, v, g, fw, fh, qx, qy, qz, zt, nx, ny, open, m);
    }
}

// ------------------------------------------------------------------------------------------
// LDS-staged depth warp.  A block owns a TY x TX tile of frame-1 pixels.  Every pixel's sample position is computed first;
// the bounding box of the block's north-west taps (+1 for the south-east ones) is the SOURCE tile of frame 2 -- the output
// tile displaced by the local flow, a few pixels larger where the flow shears or zooms.  When it fits the (TY + MY) x
// (TX + MX) staging buffers, the block computes D = m (w2z + d2 m s) and keeps m ONCE per source pixel, with coalesced row
// reads, and the four taps of every output pixel come from LDS: 2 coalesced loads per source pixel instead of 8 gathers per
// output pixel.  Otherwise (large or divergent motion) the block takes the same arithmetic from global memory as
// warp_fwd_kernel does.  Taps outside the image are staged as zeros, which is grid_sample's zeros padding; values and
// summation order per pixel are those of the gather kernels, so the results are bit-identical to them.
// Backward: the d2 gradient is accumulated in an LDS tile (ds_add_f32) and flushed with one global atomic per touched source
// pixel instead of four per output pixel.
// ------------------------------------------------------------------------------------------
// blocks of the tiled kernels that took the gather path because their source box did not fit the staging buffers
// (endo_warp_fallback_blocks: lets a test assert that a large-motion batch really exercised the fallback); [0] forward, [1] backward
// (64 counters per direction, one cache line each, a block adds to the one its index selects: with a large-motion or noisy batch most of a
// launch's 2 560 blocks take the fallback, and that many atomics on ONE address are served one after the other at ~8 ns -- DESIGN.md 4.3)
constexpr int kFallbackSlots = 64, kFallbackPad = 16;
static __device__ unsigned long long g_warp_fallback[2][kFallbackSlots][kFallbackPad] = {};
__device__ __forceinline__ void count_fallback(int which) {
    atomicAdd(&g_warp_fallback[which][(blockIdx.x + 7 * blockIdx.y + 3 * blockIdx.z) & (kFallbackSlots - 1)][0], 1ull);
}

template <int TY, int TX>
struct WarpTile {
    static constexpr int kThreads = 256;
    static constexpr int kPix = TY * TX / kThreads;          // pixels per thread
    static constexpr int MY = 8, MX = 8;                     // margin of the source tile over the output tile
    static constexpr int BH = TY + MY, BW = TX + MX;
    static_assert(TY * TX % kThreads == 0 && TX % 32 == 0, "whole pixels per thread, rows of whole half-waves");
};

// bounding box of the valid taps of a block: min / max of x0, y0 over the pixels that have a tap in range
__device__ __forceinline__ void block_tap_box(int minx, int miny, int maxx, int maxy, int* s_box, int& bx0, int& by0, int& bw, int& bh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        minx = min(minx, __shfl_down(minx, off, 64)); miny = min(miny, __shfl_down(miny, off, 64));
        maxx = max(maxx, __shfl_down(maxx, off, 64)); maxy = max(maxy, __shfl_down(maxy, off, 64));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_box[wave * 4] = minx; s_box[wave * 4 + 1] = miny; s_box[wave * 4 + 2] = maxx; s_box[wave * 4 + 3] = maxy; }
    __syncthreads();
    minx = min(min(s_box[0], s_box[4]), min(s_box[8], s_box[12]));
    miny = min(min(s_box[1], s_box[5]), min(s_box[9], s_box[13]));
    maxx = max(max(s_box[2], s_box[6]), max(s_box[10], s_box[14]));
    maxy = max(max(s_box[3], s_box[7]), max(s_box[11], s_box[15]));
    bx0 = minx; by0 = miny;
    bw = maxx - minx + 2;          // + the east / south taps
    bh = maxy - miny + 2;
}

template <int TY, int TX>
__global__ void __launch_bounds__(256) warp_fwd_tiled_kernel(const float* __restrict__ d1, const float* __restrict__ d2,
                                                             const float* __restrict__ mask, const float* __restrict__ t,
                                                             const float* __restrict__ R, const float* __restrict__ K,
                                                             float* __restrict__ warped, float* __restrict__ intersect,
                                                             int h, int w, int tiles_x, float eps) {
    using T = WarpTile<TY, TX>;
    __shared__ Camera cam;
    __shared__ int s_box[16];
    __shared__ float s_d[T::BH * T::BW], s_m[T::BH * T::BW];
    const int n = blockIdx.y;
    load_camera(K, R, t, n, &cam);
    const int64_t base = static_cast<int64_t>(n) * h * w;
    const int tx0 = (blockIdx.x % tiles_x) * TX, ty0 = (blockIdx.x / tiles_x) * TY;
    Taps tp[T::kPix];
    float mpix[T::kPix];
    int minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30);
#pragma unroll
    for (int k = 0; k < T::kPix; ++k) {
        const int idx = threadIdx.x + k * T::kThreads;
        const int yy = ty0 + idx / TX, xx = tx0 + idx % TX;
        tp[k].vw = tp[k].ve = tp[k].vn = tp[k].vs = false;
        tp[k].x0 = tp[k].y0 = 0;
        mpix[k] = 0.f;
        if (yy < h && xx < w) {
            float qx, qy, qz;
            ray(cam, static_cast<float>(xx), static_cast<float>(yy), qx, qy, qz);
            const float m = mask[base + static_cast<int64_t>(yy) * w + xx];
            const float dm = d1[base + static_cast<int64_t>(yy) * w + xx] * m;
            float zt = cam.w[2] + dm * qz;
            zt = (m > 0.5f) ? zt : eps;
            zt = (zt > 0.0f) ? zt : eps;
            tp[k] = make_taps((cam.w[0] + dm * qx) / zt, (cam.w[1] + dm * qy) / zt, w, h);
            mpix[k] = m;
            if ((tp[k].vw || tp[k].ve) && (tp[k].vn || tp[k].vs)) {
                minx = min(minx, tp[k].x0); maxx = max(maxx, tp[k].x0);
                miny = min(miny, tp[k].y0); maxy = max(maxy, tp[k].y0);
            }
        }
    }
    int bx0, by0, bw, bh;
    block_tap_box(minx, miny, maxx, maxy, s_box, bx0, by0, bw, bh);
    const bool staged = bw <= T::BW && bh <= T::BH && bw > 0;          // block-uniform; bw <= 0: no pixel of the block has a tap in range
    if (!staged && bw > 0 && threadIdx.x == 0) count_fallback(0);
    if (staged) {
        for (int e = threadIdx.x; e < bh * T::BW; e += T::kThreads) {
            const int ry = e / T::BW, rx = e - ry * T::BW;
            const int sy = by0 + ry, sx = bx0 + rx;
            float dv = 0.f, mv = 0.f, ss;
            if (rx < bw && sy >= 0 && sy < h && sx >= 0 && sx < w) dv = depth_in_1(cam, d2, mask, base, w, sx, sy, &mv, &ss);
            s_d[e] = dv; s_m[e] = mv;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < T::kPix; ++k) {
        const int idx = threadIdx.x + k * T::kThreads;
        const int yy = ty0 + idx / TX, xx = tx0 + idx % TX;
        if (yy >= h || xx >= w) continue;
        const Taps& q = tp[k];
        float v[4] = {0.f, 0.f, 0.f, 0.f}, mm[4] = {0.f, 0.f, 0.f, 0.f}, ss;
        if (staged) {
            const int o = (q.y0 - by0) * T::BW + (q.x0 - bx0);
            if (q.vw && q.vn) { v[0] = s_d[o]; mm[0] = s_m[o]; }
            if (q.ve && q.vn) { v[1] = s_d[o + 1]; mm[1] = s_m[o + 1]; }
            if (q.vw && q.vs) { v[2] = s_d[o + T::BW]; mm[2] = s_m[o + T::BW]; }
            if (q.ve && q.vs) { v[3] = s_d[o + T::BW + 1]; mm[3] = s_m[o + T::BW + 1]; }
        } else {
            if (q.vw && q.vn) v[0] = depth_in_1(cam, d2, mask, base, w, q.x0, q.y0, &mm[0], &ss);
            if (q.ve && q.vn) v[1] = depth_in_1(cam, d2, mask, base, w, q.x0 + 1, q.y0, &mm[1], &ss);
            if (q.vw && q.vs) v[2] = depth_in_1(cam, d2, mask, base, w, q.x0, q.y0 + 1, &mm[2], &ss);
            if (q.ve && q.vs) v[3] = depth_in_1(cam, d2, mask, base, w, q.x0 + 1, q.y0 + 1, &mm[3], &ss);
        }
        float acc, macc;
        tap_blend(q, v, mm, acc, macc);
        const int64_t o = base + static_cast<int64_t>(yy) * w + xx;
        warped[o] = acc;
        intersect[o] = (macc * mpix[k] >= 0.9f) ? 1.0f : 0.0f;
    }
}

template <int TY, int TX>
__global__ void __launch_bounds__(256) warp_bwd_tiled_kernel(const float* __restrict__ gw, const float* __restrict__ d1,
                                                             const float* __restrict__ d2, const float* __restrict__ mask,
                                                             const float* __restrict__ t, const float* __restrict__ R,
                                                             const float* __restrict__ K, float* __restrict__ gd1, float* gd2,
                                                             int h, int w, int tiles_x, float eps) {
    using T = WarpTile<TY, TX>;
    __shared__ Camera cam;
    __shared__ int s_box[16];
    __shared__ float s_d[T::BH * T::BW], s_m[T::BH * T::BW], s_g[T::BH * T::BW];
    const int n = blockIdx.y;
    load_camera(K, R, t, n, &cam);
    const int64_t base = static_cast<int64_t>(n) * h * w;
    const float fw = static_cast<float>(w), fh = static_cast<float>(h);
    const int tx0 = (blockIdx.x % tiles_x) * TX, ty0 = (blockIdx.x / tiles_x) * TY;
    Taps tp[T::kPix];
    float qxs[T::kPix], qys[T::kPix], qzs[T::kPix], zts[T::kPix], nxs[T::kPix], nys[T::kPix], mpix[T::kPix];
    bool opens[T::kPix];
    int minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30);
#pragma unroll
    for (int k = 0; k < T::kPix; ++k) {
        const int idx = threadIdx.x + k * T::kThreads;
        const int yy = ty0 + idx / TX, xx = tx0 + idx % TX;
        tp[k].vw = tp[k].ve = tp[k].vn = tp[k].vs = false;
        tp[k].x0 = tp[k].y0 = 0;
        mpix[k] = 0.f; opens[k] = false;
        qxs[k] = qys[k] = qzs[k] = nxs[k] = nys[k] = 0.f; zts[k] = 1.f;
        if (yy < h && xx < w) {
            ray(cam, static_cast<float>(xx), static_cast<float>(yy), qxs[k], qys[k], qzs[k]);
            const float m = mask[base + static_cast<int64_t>(yy) * w + xx];
            const float dm = d1[base + static_cast<int64_t>(yy) * w + xx] * m;
            const float z2 = cam.w[2] + dm * qzs[k];
            float zt = (m > 0.5f) ? z2 : eps;
            opens[k] = (m > 0.5f) && (zt > 0.0f);
            zt = (zt > 0.0f) ? zt : eps;
            zts[k] = zt; mpix[k] = m;
            nxs[k] = cam.w[0] + dm * qxs[k];
            nys[k] = cam.w[1] + dm * qys[k];
            tp[k] = make_taps(nxs[k] / zt, nys[k] / zt, w, h);
            if ((tp[k].vw || tp[k].ve) && (tp[k].vn || tp[k].vs)) {
                minx = min(minx, tp[k].x0); maxx = max(maxx, tp[k].x0);
                miny = min(miny, tp[k].y0); maxy = max(maxy, tp[k].y0);
            }
        }
    }
    int bx0, by0, bw, bh;
    block_tap_box(minx, miny, maxx, maxy, s_box, bx0, by0, bw, bh);
    const bool staged = bw <= T::BW && bh <= T::BH && bw > 0;
    if (!staged && bw > 0 && threadIdx.x == 0) count_fallback(1);
    if (staged) {
        for (int e = threadIdx.x; e < bh * T::BW; e += T::kThreads) {
            const int ry = e / T::BW, rx = e - ry * T::BW;
            const int sy = by0 + ry, sx = bx0 + rx;
            float dv = 0.f, mv = 0.f, ss;
            if (rx < bw && sy >= 0 && sy < h && sx >= 0 && sx < w) dv = depth_in_1(cam, d2, mask, base, w, sx, sy, &mv, &ss);
            s_d[e] = dv; s_m[e] = mv; s_g[e] = 0.f;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < T::kPix; ++k) {
        const int idx = threadIdx.x + k * T::kThreads;
        const int yy = ty0 + idx / TX, xx = tx0 + idx % TX;
        if (yy >= h || xx >= w) continue;
        const Taps& q = tp[k];
        const float g = gw[base + static_cast<int64_t>(yy) * w + xx];
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const bool val[4] = {q.vw && q.vn, q.ve && q.vn, q.vw && q.vs, q.ve && q.vs};
        const float wt[4] = {q.wnw, q.wne, q.wsw, q.wse};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (!val[c]) continue;
            const int sx = q.x0 + (c & 1), sy = q.y0 + (c >> 1);
            float mm, ss;
            if (staged) {
                const int o = (sy - by0) * T::BW + (sx - bx0);
                v[c] = s_d[o]; mm = s_m[o];
                ss = plane_s(cam, sx, sy);
                atomicAdd(&s_g[o], warp_grad_d2_term(g, wt[c], mm, ss));
            } else {
                v[c] = depth_in_1(cam, d2, mask, base, w, sx, sy, &mm, &ss);
                atomicAdd(gd2 + base + static_cast<int64_t>(sy) * w + sx, warp_grad_d2_term(g, wt[c], mm, ss));
            }
        }
        gd1[base + static_cast<int64_t>(yy) * w + xx] = warp_grad_d1(q, v, g, fw, fh, qxs[k], qys[k], qzs[k], zts[k], nxs[k], nys[k], opens[k], mpix[k]);
    }
    if (staged) {
        __syncthreads();
        for (int e = threadIdx.x; e < bh * T::BW; e += T::kThreads) {
            const float gsum = s_g[e];
            if (gsum != 0.f) {
                const int ry = e / T::BW, rx = e - ry * T::BW;
                atomicAdd(gd2 + base + static_cast<int64_t>(by0 + ry) * w + bx0 + rx, gsum);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Depth warp both ways + depth-consistency loss (losses.py:112-146), forward and backward, as TWO kernels (+ a 1 KB memset and a one-wave finalize):
// the chain of BASELINE's second metric (endo_warp_consistency, train.py:305-314).  Round 3 composed it from the entry points above:
// 11 launches, ~100 us per call for 105 MB.  Here gridDim.z = 2 covers the two directions (z = 0: frame 1 warped from frame 2,
// pose 1-wrt-2; z = 1: the roles swapped):
//   consistency_fwd_kernel   the tiled warp forward; per pixel the four NormalizedDistanceLoss sums of its sample (block reduction,
//                            one fp64 atomic per sum and block); it also ZEROES its tile of this direction's own gradient output.
//   consistency_finalize_kernel  one wave: the 2 x n x 4 sums -> the loss and the per-sample coefficients (d loss / d numerator,
//                            d loss / d denominator) the backward kernel needs.
//   consistency_bwd_kernel   per pixel the loss's gradient w.r.t. its own depth and w.r.t. the warped depth, the latter pushed
//                            straight through the warp backward of the same pixel (taps recomputed, source tile staged as in the
//                            forward kernel): its own-pixel terms are ONE atomic add into this direction's gradient, the source terms
//                            go through the LDS tile into the other direction's gradient.
// Same arithmetic per term as the kernels above; the three contributions of a gradient element arrive by fp32 atomics in any order
// (the composed path added them in a fixed order: differences at fp32 rounding, test bound 1e-6).
// ------------------------------------------------------------------------------------------
// One cache line per (direction, sample) for the forward kernel's four sums: its 2 560 blocks add them with one 32-byte atomic each, and
// atomics on one line are served one after the other (~8 ns) -- packed 4 doubles apart, four samples shared a line.
constexpr int kStatStride = 16;          // doubles

struct ConsistencyArgs {
    const float* depth[2];           // [0] frame 1, [1] frame 2
    const float* t[2];               // [0] 1-wrt-2, [1] 2-wrt-1
    const float* R[2];
    const float* K;
    const float* mask;
    float* warped[2];                // direction z: depth[1 - z] warped into frame z
    float* inter[2];
    float* grad[2];                  // d loss / d depth[z]
    double* stats;                   // [2][n][kStatStride], the first 4: sum m a, sum m, sum m |P - Pw|_1, sum m (a + |b|)   (zeroed by the caller)
    float* coef;                     // [2][n][2]  (d loss / d numerator, d loss / d denominator) of sample n in direction z
    float* loss;
    float c_dcl;                     // dcl_weight * 0.5
    float eps;
    int n, h, w, tiles_x;
    int zero_grads;                  // 1: the forward kernel zeroes grad[]; 0: the caller has initialised them (endo_loss_head: the flow terms)
    int loss_in_bwd;                 // 1: no finalize kernel ran; the backward kernel's first block writes the loss
};

__device__ __forceinline__ float sgnf(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

template <int TY, int TX>
__global__ void __launch_bounds__(256) consistency_fwd_kernel(const ConsistencyArgs a) {
    using T = WarpTile<TY, TX>;
    __shared__ Camera cam;
    __shared__ int s_box[16];
    __shared__ float s_d[T::BH * T::BW], s_m[T::BH * T::BW];
    __shared__ double scratch[4 * 4];
    const int n = blockIdx.y, z = blockIdx.z;
    const int h = a.h, w = a.w;
    const float* __restrict__ d1 = a.depth[z];
    const float* __restrict__ d2 = a.depth[1 - z];
    const float* __restrict__ mask = a.mask;
    load_camera(a.K, a.R[z], a.t[z], n, &cam);
    const int64_t base = static_cast<int64_t>(n) * h * w;
    const int tx0 = (blockIdx.x % a.tiles_x) * TX, ty0 = (blockIdx.x / a.tiles_x) * TY;
    Taps tp[T::kPix];
    float mpix[T::kPix], dself[T::kPix];
    int minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30);
#pragma unroll
    for (int k = 0; k < T::kPix; ++k) {
        const int idx = threadIdx.x + k * T::kThreads;
        const int yy = ty0 + idx / TX, xx = tx0 + idx % TX;
        tp[k].vw = tp[k].ve = tp[k].vn = tp[k].vs = false;
        tp[k].x0 = tp[k].y0 = 0;
        mpix[k] = 0.f; dself[k] = 0.f;
        if (yy < h && xx < w) {
            float qx, qy, qz;
            ray(cam, static_cast<float>(xx), static_cast<float>(yy), qx, qy, qz);
            const float m = mask[base + static_cast<int64_t>(yy) * w + xx];
            dself[k] = d1[base + static_cast<int64_t>(yy) * w + xx];
            const float dm = dself[k] * m;
            float zt = cam.w[2] + dm * qz;
            zt = (m > 0.5f) ? zt : a.eps;
            zt = (zt > 0.0f) ? zt : a.eps;
            tp[k] = make_taps((cam.w[0] + dm * qx) / zt, (cam.w[1] + dm * qy) / zt, w, h);
            mpix[k] = m;
            if ((tp[k].vw || tp[k].ve) && (tp[k].vn || tp[k].vs)) {
                minx = min(minx, tp[k].x0); maxx = max(maxx, tp[k].x0);
                miny = min(miny, tp[k].y0); maxy = max(maxy, tp[k].y0);
            }
        }
    }
    int bx0, by0, bw, bh;
    block_tap_box(minx, miny, maxx, maxy, s_box, bx0, by0, bw, bh);
    const bool staged = bw <= T::BW && bh <= T::BH && bw > 0;
    if (!staged && bw > 0 && threadIdx.x == 0) count_fallback(0);
    if (staged) {
        for (int e = threadIdx.x; e < bh * T::BW; e += T::kThreads) {
            const int ry = e / T::BW, rx = e - ry * T::BW;
            const int sy = by0 + ry, sx = bx0 + rx;
            float dv = 0.f, mv = 0.f, ss;
            if (rx < bw && sy >= 0 && sy < h && sx >= 0 && sx < w) dv = depth_in_1(cam, d2, mask, base, w, sx, sy, &mv, &ss);
            s_d[e] = dv; s_m[e] = mv;
        }
        __syncthreads();
    }
    const float fx = a.K[9 * n + 0], fy = a.K[9 * n + 4], cx = a.K[9 * n + 2], cy = a.K[9 * n + 5];
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < T::kPix; ++k) {
        const int idx = threadIdx.x + k * T::kThreads;
        const int yy = ty0 + idx / TX, xx = tx0 + idx % TX;
        if (yy >= h || xx >= w) continue;
        const Taps& q = tp[k];
        float v[4] = {0.f, 0.f, 0.f, 0.f}, mm[4] = {0.f, 0.f, 0.f, 0.f}, ss;
        if (staged) {
            const int o = (q.y0 - by0) * T::BW + (q.x0 - bx0);
            if (q.vw && q.vn) { v[0] = s_d[o]; mm[0] = s_m[o]; }
            if (q.ve && q.vn) { v[1] = s_d[o + 1]; mm[1] = s_m[o + 1]; }
            if (q.vw && q.vs) { v[2] = s_d[o + T::BW]; mm[2] = s_m[o + T::BW]; }
            if (q.ve && q.vs) { v[3] = s_d[o + T::BW + 1]; mm[3] = s_m[o + T::BW + 1]; }
        } else {
            if (q.vw && q.vn) v[0] = depth_in_1(cam, d2, mask, base, w, q.x0, q.y0, &mm[0], &ss);
            if (q.ve && q.vn) v[1] = depth_in_1(cam, d2, mask, base, w, q.x0 + 1, q.y0, &mm[1], &ss);
            if (q.vw && q.vs) v[2] = depth_in_1(cam, d2, mask, base, w, q.x0, q.y0 + 1, &mm[2], &ss);
            if (q.ve && q.vs) v[3] = depth_in_1(cam, d2, mask, base, w, q.x0 + 1, q.y0 + 1, &mm[3], &ss);
        }
        float acc, macc;
        tap_blend(q, v, mm, acc, macc);
        const int64_t o = base + static_cast<int64_t>(yy) * w + xx;
        const float mi = (macc * mpix[k] >= 0.9f) ? 1.0f : 0.0f;
        a.warped[z][o] = acc;
        a.inter[z][o] = mi;
        if (a.zero_grads) a.grad[z][o] = 0.f;          // the backward kernel accumulates into it
        // NormalizedDistanceLoss sums (norm_dist_reduce, losses.hip): a = own depth, b = warped depth
        const float ax = (static_cast<float>(xx) - cx) / fx, ay = (static_cast<float>(yy) - cy) / fy;
        const float av = dself[k], b = acc;
        part[0] += mi * av;
        part[1] += mi;
        part[2] += mi * fabsf(ax * av - ax * b) + mi * fabsf(ay * av - ay * b) + mi * fabsf(av - b);
        part[3] += mi * (av + fabsf(b));
    }
    block_sum_atomic<4>(part, a.stats + kStatStride * (z * a.n + n), scratch);
}

// loss and the backward pass's per-sample coefficients from the 2 x n x 4 sums.  A kernel of its own, one wave: the "last block of the
// forward kernel finalises" form needs a device-scope fence in every block, which on this part writes back and invalidates the XCD's L2
// -- 2560 blocks of it made the forward kernel 222 us instead of ~15 (profiles/r04_warp_consistency_kernel_stats.txt history).
__global__ void consistency_finalize_kernel(const ConsistencyArgs a) {
    // lane i < 2 n owns (direction, sample) i: its term and coefficients; a wave reduction adds the terms (n <= 32; larger batches loop)
    float term = 0.f;
    for (int i = threadIdx.x; i < 2 * a.n; i += 64) {
        const double* st = a.stats + kStatStride * i;
        const float s0 = static_cast<float>(st[0]), s1 = static_cast<float>(st[1]), s2 = static_cast<float>(st[2]), s3 = static_cast<float>(st[3]);
        const float mean_value = s0 / (1.0e-5f + s1);                  // norm_dist_den (losses.hip) with the module's eps
        const float den = 1.0e-5f * mean_value + s3;
        term += 2.0f * s2 / den;
        const float g = a.c_dcl / static_cast<float>(a.n);             // d loss / d (this sample's term)
        a.coef[2 * i] = 2.0f * g / den;
        a.coef[2 * i + 1] = -2.0f * g * s2 / (den * den);
    }
    term = wave_sum(term);
    if (threadIdx.x == 0) a.loss[0] = a.c_dcl * (term / static_cast<float>(a.n));
}

template <int TY, int TX>
__global__ void __launch_bounds__(256) consistency_bwd_kernel(const ConsistencyArgs a) {
    using T = WarpTile<TY, TX>;
    __shared__ Camera cam;
    __shared__ int s_box[16];
    __shared__ float s_d[T::BH * T::BW], s_m[T::BH * T::BW], s_g[T::BH * T::BW];
    const int n = blockIdx.y, z = blockIdx.z;
    const int h = a.h, w = a.w;
    const float* __restrict__ d1 = a.depth[z];
    const float* __restrict__ d2 = a.depth[1 - z];
    const float* __restrict__ mask = a.mask;
    float* gd1 = a.grad[z];
    float* gd2 = a.grad[1 - z];
    load_camera(a.K, a.R[z], a.t[z], n, &cam);
    const int64_t base = static_cast<int64_t>(n) * h * w;
    const float fw = static_cast<float>(w), fh = static_cast<float>(h);
    const int tx0 = (blockIdx.x % a.tiles_x) * TX, ty0 = (blockIdx.x / a.tiles_x) * TY;
    // this sample's coefficients from the forward kernel's sums (consistency_finalize_kernel's arithmetic; the stand-alone call skips
    // that kernel and block (0, 0, 0) writes the loss here)
    float cnum, cden;
    {
        const double* st = a.stats + kStatStride * (z * a.n + n);
        const float s0 = static_cast<float>(st[0]), s1 = static_cast<float>(st[1]), s2 = static_cast<float>(st[2]), s3 = static_cast<float>(st[3]);
        const float den = 1.0e-5f * (s0 / (1.0e-5f + s1)) + s3;
        const float g = a.c_dcl / static_cast<float>(a.n);
        cnum = 2.0f * g / den;
        cden = -2.0f * g * s2 / (den * den);
    }
    if (a.loss_in_bwd && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 64) {
        float term = 0.f;
        for (int i = threadIdx.x; i < 2 * a.n; i += 64) {
            const double* st = a.stats + kStatStride * i;
            const float s0 = static_cast<float>(st[0]), s1 = static_cast<float>(st[1]), s2 = static_cast<float>(st[2]), s3 = static_cast<float>(st[3]);
            term += 2.0f * s2 / (1.0e-5f * (s0 / (1.0e-5f + s1)) + s3);
        }
        term = wave_sum(term);
        if (threadIdx.x == 0) a.loss[0] = a.c_dcl * (term / static_cast<float>(a.n));
    }
    const float fx = a.K[9 * n + 0], fy = a.K[9 * n + 4], cx = a.K[9 * n + 2], cy = a.K[9 * n + 5];
    Taps tp[T::kPix];
    float qxs[T::kPix], qys[T::kPix], qzs[T::kPix], zts[T::kPix], nxs[T::kPix], nys[T::kPix], mpix[T::kPix], dself[T::kPix];
    bool opens[T::kPix];
    int minx = 1 << 30, miny = 1 << 30, maxx = -(1 << 30), maxy = -(1 << 30);
#pragma unroll
    for (int k = 0; k < T::kPix; ++k) {
        const int idx = threadIdx.x + k * T::kThreads;
        const int yy = ty0 + idx / TX, xx = tx0 + idx % TX;
        tp[k].vw = tp[k].ve = tp[k].vn = tp[k].vs = false;
        tp[k].x0 = tp[k].y0 = 0;
        mpix[k] = 0.f; opens[k] = false; dself[k] = 0.f;
        qxs[k] = qys[k] = qzs[k] = nxs[k] = nys[k] = 0.f; zts[k] = 1.f;
        if (yy < h && xx < w) {
            ray(cam, static_cast<float>(xx), static_cast<float>(yy), qxs[k], qys[k], qzs[k]);
            const float m = mask[base + static_cast<int64_t>(yy) * w + xx];
            dself[k] = d1[base + static_cast<int64_t>(yy) * w + xx];
            const float dm = dself[k] * m;
            const float z2 = cam.w[2] + dm * qzs[k];
            float zt = (m > 0.5f) ? z2 : a.eps;
            opens[k] = (m > 0.5f) && (zt > 0.0f);
            zt = (zt > 0.0f) ? zt : a.eps;
            zts[k] = zt; mpix[k] = m;
            nxs[k] = cam.w[0] + dm * qxs[k];
            nys[k] = cam.w[1] + dm * qys[k];
            tp[k] = make_taps(nxs[k] / zt, nys[k] / zt, w, h);
            if ((tp[k].vw || tp[k].ve) && (tp[k].vn || tp[k].vs)) {
                minx = min(minx, tp[k].x0); maxx = max(maxx, tp[k].x0);
                miny = min(miny, tp[k].y0); maxy = max(maxy, tp[k].y0);
            }
        }
    }
    int bx0, by0, bw, bh;
    block_tap_box(minx, miny, maxx, maxy, s_box, bx0, by0, bw, bh);
    const bool staged = bw <= T::BW && bh <= T::BH && bw > 0;
    if (!staged && bw > 0 && threadIdx.x == 0) count_fallback(1);
    if (staged) {
        for (int e = threadIdx.x; e < bh * T::BW; e += T::kThreads) {
            const int ry = e / T::BW, rx = e - ry * T::BW;
            const int sy = by0 + ry, sx = bx0 + rx;
            float dv = 0.f, mv = 0.f, ss;
            if (rx < bw && sy >= 0 && sy < h && sx >= 0 && sx < w) dv = depth_in_1(cam, d2, mask, base, w, sx, sy, &mv, &ss);
            s_d[e] = dv; s_m[e] = mv; s_g[e] = 0.f;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < T::kPix; ++k) {
        const int idx = threadIdx.x + k * T::kThreads;
        const int yy = ty0 + idx / TX, xx = tx0 + idx % TX;
        if (yy >= h || xx >= w) continue;
        const Taps& q = tp[k];
        const int64_t o = base + static_cast<int64_t>(yy) * w + xx;
        // the loss's gradient at this pixel (norm_dist_bwd_kernel, losses.hip): a = own depth, b = warped depth, m = intersection mask
        const float ax = (static_cast<float>(xx) - cx) / fx, ay = (static_cast<float>(yy) - cy) / fy;
        const float mi = a.inter[z][o], av = dself[k], b = a.warped[z][o];
        const float tt = ax * sgnf(ax * av - ax * b) + ay * sgnf(ay * av - ay * b) + sgnf(av - b);
        const float g_own = mi * (cnum * tt + cden);
        const float g = mi * (-cnum * tt + cden * sgnf(b));          // d loss / d warped: the warp backward's incoming gradient
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const bool val[4] = {q.vw && q.vn, q.ve && q.vn, q.vw && q.vs, q.ve && q.vs};
        const float wt[4] = {q.wnw, q.wne, q.wsw, q.wse};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (!val[c]) continue;
            const int sx = q.x0 + (c & 1), sy = q.y0 + (c >> 1);
            float mm, ss;
            if (staged) {
                const int oo = (sy - by0) * T::BW + (sx - bx0);
                v[c] = s_d[oo]; mm = s_m[oo];
                ss = plane_s(cam, sx, sy);
                atomicAdd(&s_g[oo], warp_grad_d2_term(g, wt[c], mm, ss));
            } else {
                v[c] = depth_in_1(cam, d2, mask, base, w, sx, sy, &mm, &ss);
                atomicAdd(gd2 + base + static_cast<int64_t>(sy) * w + sx, warp_grad_d2_term(g, wt[c], mm, ss));
            }
        }
        atomicAdd(gd1 + o, g_own + warp_grad_d1(q, v, g, fw, fh, qxs[k], qys[k], qzs[k], zts[k], nxs[k], nys[k], opens[k], mpix[k]));
    }
    if (staged) {
        __syncthreads();
        for (int e = threadIdx.x; e < bh * T::BW; e += T::kThreads) {
            const float gsum = s_g[e];
            if (gsum != 0.f) {
                const int ry = e / T::BW, rx = e - ry * T::BW;
                atomicAdd(gd2 + base + static_cast<int64_t>(by0 + ry) * w + bx0 + rx, gsum);
            }
        }
    }
}

__global__ void __launch_bounds__(256) mask_mul_kernel(const float* __restrict__ a, const float* __restrict__ mask,
                                                       float* __restrict__ out, int c, int hw) {
    const int n = blockIdx.y;
    const int64_t mbase = static_cast<int64_t>(n) * hw;
    const int64_t abase = mbase * c;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const float m = mask[mbase + i];
        for (int k = 0; k < c; ++k) out[abase + static_cast<int64_t>(k) * hw + i] = a[abase + static_cast<int64_t>(k) * hw + i] * m;
    }
}

// tile of the default entry points endo_depth_warp_fwd / _bwd (profiles/r02_warp_tile_sweep.txt); 0 x 0 = the gather kernels
constexpr int kWarpTileH = 16, kWarpTileW = 32;

inline int plane_blocks(int hw, int threads) {
    int b = (hw + threads - 1) / threads;
    return b < 1 ? 1 : (b > 1024 ? 1024 : b);
}

}  // namespace endo

using namespace endo;

extern "C" int endo_depth_scale_fwd(const float* pred, const float* sparse_depth, const float* sparse_mask, float* scaled,
                                    float* ratio, double* stats, int n, int hw, float eps, void* stream_) {
    return endo_depth_scale_fwd_impl(pred, sparse_depth, sparse_mask, scaled, ratio, stats, n, hw, eps, 1, static_cast<hipStream_t>(stream_));
}

// zero = 0: the caller has zeroed `stats` (the loss head: one memset for all its reduction tables)
int endo_depth_scale_fwd_impl(const float* pred, const float* sparse_depth, const float* sparse_mask, float* scaled, float* ratio, double* stats,
                              int n, int hw, float eps, int zero, hipStream_t stream) {
    if (!pred || !sparse_depth || !sparse_mask || !scaled || !ratio || !stats || n <= 0 || hw <= 0) return ENDO_E_BADARG;
    ProfScope prof(kProfGeometry, stream, 0.0, 6.0 * 4.0 * n * hw);
    if (zero) ENDO_CHECK(hipMemsetAsync(stats, 0, sizeof(double) * 8 * n, stream));
    dim3 rgrid((hw + kRedThreads * kRedItems - 1) / (kRedThreads * kRedItems), n);
    depth_scale_pass1<<<rgrid, kRedThreads, 0, stream>>>(sparse_depth, sparse_mask, stats, hw);
    depth_scale_pass2<<<rgrid, kRedThreads, 0, stream>>>(pred, sparse_depth, stats, hw, eps);
    depth_scale_finalize<<<1, 64, 0, stream>>>(stats, ratio, n);
    depth_scale_apply<<<dim3(plane_blocks(hw, 256), n), 256, 0, stream>>>(pred, stats, scaled, hw);
    ENDO_LAUNCH_CHECK();
    return 0;
}

extern "C" int endo_depth_scale_bwd(const float* grad_scaled, const float* grad_ratio, const float* pred,
                                    const float* sparse_depth, const double* stats, float* grad_pred, double* work, int n,
                                    int hw, float eps, void* stream_) {
    return endo_depth_scale_bwd_impl(grad_scaled, grad_ratio, pred, sparse_depth, stats, grad_pred, work, n, hw, eps, 1, static_cast<hipStream_t>(stream_));
}

int endo_depth_scale_bwd_impl(const float* grad_scaled, const float* grad_ratio, const float* pred, const float* sparse_depth, const double* stats,
                              float* grad_pred, double* work, int n, int hw, float eps, int zero, hipStream_t stream) {
    if (!pred || !sparse_depth || !stats || !grad_pred || !work || n <= 0 || hw <= 0) return ENDO_E_BADARG;
    ProfScope prof(kProfGeometry, stream, 0.0, 6.0 * 4.0 * n * hw);
    if (zero) ENDO_CHECK(hipMemsetAsync(work, 0, sizeof(double) * n, stream));
    if (grad_scaled) {
        dim3 rgrid((hw + kRedThreads * kRedItems - 1) / (kRedThreads * kRedItems), n);
        depth_scale_bwd_reduce<<<rgrid, kRedThreads, 0, stream>>>(grad_scaled, pred, work, hw);
    }
    depth_scale_bwd_apply<<<dim3(plane_blocks(hw, 256), n), 256, 0, stream>>>(grad_scaled, grad_ratio, pred, sparse_depth,
                                                                               stats, work, grad_pred, n, hw, eps);
    ENDO_LAUNCH_CHECK();
    return 0;
}

extern "C" int endo_flow_from_depth_fwd(const float* depth, const float* mask, const float* t, const float* R, const float* K,
                                        float* flow, int n, int h, int w, void* stream_) {
    if (!depth || !mask || !t || !R || !K || !flow || n <= 0 || h <= 0 || w <= 0) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(kProfGeometry, stream, 0.0, 4.0 * 4.0 * n * h * w);
    flow_fwd_kernel<<<dim3(plane_blocks(h * w, 256), n), 256, 0, stream>>>(depth, mask, t, R, K, flow, h, w);
    ENDO_LAUNCH_CHECK();
    return 0;
}

extern "C" int endo_flow_from_depth_bwd(const float* grad_flow, const float* depth, const float* mask, const float* t,
                                        const float* R, const float* K, float* grad_depth, int n, int h, int w, void* stream_) {
    if (!grad_flow || !depth || !mask || !t || !R || !K || !grad_depth || n <= 0 || h <= 0 || w <= 0) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(kProfGeometry, stream, 0.0, 5.0 * 4.0 * n * h * w);
    flow_bwd_kernel<<<dim3(plane_blocks(h * w, 256), n), 256, 0, stream>>>(grad_flow, depth, mask, t, R, K, grad_depth, h, w);
    ENDO_LAUNCH_CHECK();
    return 0;
}

template <int TY, int TX>
static int launch_warp_fwd_tiled(const float* d1, const float* d2, const float* mask, const float* t, const float* R, const float* K,
                                 float* warped, float* intersect, int n, int h, int w, float eps, hipStream_t stream) {
    const int tiles_x = (w + TX - 1) / TX, tiles_y = (h + TY - 1) / TY;
    warp_fwd_tiled_kernel<TY, TX><<<dim3(tiles_x * tiles_y, n), 256, 0, stream>>>(d1, d2, mask, t, R, K, warped, intersect, h, w, tiles_x, eps);
    return static_cast<int>(hipGetLastError());
}

template <int TY, int TX>
static int launch_warp_bwd_tiled(const float* gw, const float* d1, const float* d2, const float* mask, const float* t, const float* R,
                                 const float* K, float* gd1, float* gd2, int n, int h, int w, float eps, hipStream_t stream) {
    const int tiles_x = (w + TX - 1) / TX, tiles_y = (h + TY - 1) / TY;
    warp_bwd_tiled_kernel<TY, TX><<<dim3(tiles_x * tiles_y, n), 256, 0, stream>>>(gw, d1, d2, mask, t, R, K, gd1, gd2, h, w, tiles_x, eps);
    return static_cast<int>(hipGetLastError());
}

extern "C" int endo_warp_fallback_blocks(long long* forward, long long* backward, int reset) {
    static unsigned long long host[2][kFallbackSlots][kFallbackPad];
    ENDO_CHECK(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_warp_fallback), sizeof(host)));          // synchronises with the device: a test hook
    long long total[2] = {0, 0};
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < kFallbackSlots; ++i) total[k] += static_cast<long long>(host[k][i][0]);
    if (forward) *forward = total[0];
    if (backward) *backward = total[1];
    if (reset) {
        static const unsigned long long zero[2][kFallbackSlots][kFallbackPad] = {};
        ENDO_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_warp_fallback), zero, sizeof(zero)));
    }
    return 0;
}

extern "C" int endo_depth_warp_fwd_tiled(const float* depth_1, const float* depth_2, const float* mask, const float* t, const float* R,
                                         const float* K, float* warped, float* intersect, int n, int h, int w, float eps,
                                         int tile_h, int tile_w, void* stream_) {
    if (!depth_1 || !depth_2 || !mask || !t || !R || !K || !warped || !intersect || n <= 0 || h <= 0 || w <= 0)
        return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(kProfGeometry, stream, 0.0, 5.0 * 4.0 * n * h * w);
    if (tile_h == 0 && tile_w == 0) {
        warp_fwd_kernel<<<dim3(plane_blocks(h * w, 256), n), 256, 0, stream>>>(depth_1, depth_2, mask, t, R, K, warped, intersect,
                                                                               h, w, eps);
        ENDO_LAUNCH_CHECK();
        return 0;
    }
#define ENDO_WARP_FWD(TY_, TX_) if (tile_h == TY_ && tile_w == TX_) return launch_warp_fwd_tiled<TY_, TX_>(depth_1, depth_2, mask, t, R, K, warped, intersect, n, h, w, eps, stream)
    ENDO_WARP_FWD(8, 32); ENDO_WARP_FWD(16, 32); ENDO_WARP_FWD(16, 64); ENDO_WARP_FWD(32, 32); ENDO_WARP_FWD(32, 64);
#undef ENDO_WARP_FWD
    return ENDO_E_UNSUPPORTED;
}

extern "C" int endo_depth_warp_fwd(const float* depth_1, const float* depth_2, const float* mask, const float* t, const float* R,
                                   const float* K, float* warped, float* intersect, int n, int h, int w, float eps,
                                   void* stream_) {
    return endo_depth_warp_fwd_tiled(depth_1, depth_2, mask, t, R, K, warped, intersect, n, h, w, eps, kWarpTileH, kWarpTileW, stream_);
}

extern "C" int endo_depth_warp_bwd_tiled(const float* grad_warped, const float* depth_1, const float* depth_2, const float* mask,
                                         const float* t, const float* R, const float* K, float* grad_d1, float* grad_d2, int n,
                                         int h, int w, float eps, int tile_h, int tile_w, void* stream_) {
    if (!grad_warped || !depth_1 || !depth_2 || !mask || !t || !R || !K || !grad_d1 || !grad_d2 || n <= 0 || h <= 0 || w <= 0)
        return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(kProfGeometry, stream, 0.0, 7.0 * 4.0 * n * h * w);
    ENDO_CHECK(hipMemsetAsync(grad_d2, 0, sizeof(float) * static_cast<size_t>(n) * h * w, stream));
    if (tile_h == 0 && tile_w == 0) {
        warp_bwd_kernel<<<dim3(plane_blocks(h * w, 256), n), 256, 0, stream>>>(grad_warped, depth_1, depth_2, mask, t, R, K,
                                                                               grad_d1, grad_d2, h, w, eps);
        ENDO_LAUNCH_CHECK();
        return 0;
    }
#define ENDO_WARP_BWD(TY_, TX_) if (tile_h == TY_ && tile_w == TX_) return launch_warp_bwd_tiled<TY_, TX_>(grad_warped, depth_1, depth_2, mask, t, R, K, grad_d1, grad_d2, n, h, w, eps, stream)
    ENDO_WARP_BWD(8, 32); ENDO_WARP_BWD(16, 32); ENDO_WARP_BWD(16, 64); ENDO_WARP_BWD(32, 32); ENDO_WARP_BWD(32, 64);
#undef ENDO_WARP_BWD
    return ENDO_E_UNSUPPORTED;
}

extern "C" int endo_depth_warp_bwd(const float* grad_warped, const float* depth_1, const float* depth_2, const float* mask,
                                   const float* t, const float* R, const float* K, float* grad_d1, float* grad_d2, int n,
                                   int h, int w, float eps, void* stream_) {
    return endo_depth_warp_bwd_tiled(grad_warped, depth_1, depth_2, mask, t, R, K, grad_d1, grad_d2, n, h, w, eps, kWarpTileH, kWarpTileW,
                                     stream_);
}

extern "C" int endo_mask_mul(const float* a, const float* mask, float* out, int n, int c, int hw, void* stream_) {
    if (!a || !mask || !out || n <= 0 || c <= 0 || hw <= 0) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    mask_mul_kernel<<<dim3(plane_blocks(hw, 256), n), 256, 0, stream>>>(a, mask, out, c, hw);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// Depth warp both ways + depth-consistency loss, forward and backward (include/endo_hip.h): one small memset and two kernels.
extern "C" int64_t endo_warp_consistency_workspace_floats(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0) return -1;
    const int64_t p = static_cast<int64_t>(n) * h * w;
    return 4 * (p + 3) + (4 * kStatStride + 16) * n + 64;
}

// Algorithmic HBM bytes of one endo_warp_consistency call, as SURVEY.md 8(d) counts the chain per pixel and frame pair, both directions:
// forward 2 x (read depth_1, depth_2, boundary; 4-tap gather of the source depth; write warped depth + intersect mask) = 2 x 36 B,
// backward 2 x (read the two maps, the mask, the forward's warped / intersect planes; write d loss / d depth of the target; 4-tap
// scatter into the source's gradient) = 2 x 44 B: 160 B per pixel of a pair.  bench.py's roofline_depth_warp divides by the device time.
extern "C" int64_t endo_warp_consistency_bytes(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0) return -1;
    return static_cast<int64_t>(160) * n * h * w;
}

// phase 1: memset + forward kernel (loss, coefficients, and -- zero_grads -- cleared gradients); phase 2: backward kernel (atomic adds
// into grad_depth_*).  endo_warp_consistency runs both; endo_loss_head (head.hip) runs them around its other terms.
int endo_consistency_phase(int phase, const float* depth_1, const float* depth_2, const float* boundaries, const float* t_1_wrt_2,
                           const float* r_1_wrt_2, const float* t_2_wrt_1, const float* r_2_wrt_1, const float* intrinsics, float dcl_weight,
                           float eps, float* loss, float* grad_depth_1, float* grad_depth_2, float* workspace, int n, int h, int w,
                           int zero_grads, hipStream_t stream) {
    const int64_t p = static_cast<int64_t>(n) * h * w;
    float* ws = workspace;
    auto take = [&](int64_t count) { float* q = ws; ws += (count + 3) / 4 * 4; return q; };
    ConsistencyArgs a{};
    a.depth[0] = depth_1; a.depth[1] = depth_2;
    a.t[0] = t_1_wrt_2; a.t[1] = t_2_wrt_1;
    a.R[0] = r_1_wrt_2; a.R[1] = r_2_wrt_1;
    a.K = intrinsics; a.mask = boundaries;
    a.warped[0] = take(p); a.warped[1] = take(p);
    a.inter[0] = take(p); a.inter[1] = take(p);
    a.grad[0] = grad_depth_1; a.grad[1] = grad_depth_2;
    float* zeroed = take(2 * 2 * kStatStride * n);          // the sums (doubles), zeroed per call; the coefficients are written by the finalize kernel
    a.stats = reinterpret_cast<double*>(zeroed);
    a.coef = take(4 * n);
    a.loss = loss;
    a.c_dcl = static_cast<float>(static_cast<double>(dcl_weight) * 0.5);
    a.eps = eps;
    a.n = n; a.h = h; a.w = w;
    a.zero_grads = zero_grads;
    a.loss_in_bwd = zero_grads;          // the stand-alone call (endo_warp_consistency) needs the loss only at its end
    constexpr int TY = kWarpTileH, TX = kWarpTileW;
    a.tiles_x = (w + TX - 1) / TX;
    const int tiles_y = (h + TY - 1) / TY;
    const dim3 grid(a.tiles_x * tiles_y, n, 2);
    if (phase == 1) {
        ENDO_CHECK(hipMemsetAsync(zeroed, 0, sizeof(float) * (2 * 2 * kStatStride * n), stream));
        consistency_fwd_kernel<TY, TX><<<grid, 256, 0, stream>>>(a);
        if (!a.loss_in_bwd) consistency_finalize_kernel<<<1, 64, 0, stream>>>(a);          // the loss head reads the loss between the phases
    } else {
        consistency_bwd_kernel<TY, TX><<<grid, 256, 0, stream>>>(a);
    }
    ENDO_LAUNCH_CHECK();
    return 0;
}

extern "C" int endo_warp_consistency(const float* depth_1, const float* depth_2, const float* boundaries, const float* t_1_wrt_2,
                                     const float* r_1_wrt_2, const float* t_2_wrt_1, const float* r_2_wrt_1, const float* intrinsics,
                                     float dcl_weight, float eps, float* loss, float* grad_depth_1, float* grad_depth_2,
                                     float* workspace, int n, int h, int w, void* stream_) {
    if (!depth_1 || !depth_2 || !boundaries || !t_1_wrt_2 || !r_1_wrt_2 || !t_2_wrt_1 || !r_2_wrt_1 || !intrinsics || !loss ||
        !grad_depth_1 || !grad_depth_2 || !workspace || n <= 0 || h <= 0 || w <= 0)
        return ENDO_E_BADARG;
    if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // algorithmic bytes as SURVEY.md 8(d) counts the chain: per direction warp fwd 5 planes + bwd 7, loss fwd 3 + bwd 5 = 20 planes = 160 B per pixel
    ProfScope prof(kProfGeometry, stream, 0.0, 160.0 * static_cast<double>(n) * h * w);
    for (int phase = 1; phase <= 2; ++phase) {
        const int rc = endo_consistency_phase(phase, depth_1, depth_2, boundaries, t_1_wrt_2, r_1_wrt_2, t_2_wrt_1, r_2_wrt_1, intrinsics,
                                              dcl_weight, eps, loss, grad_depth_1, grad_depth_2, workspace, n, h, w, 1, stream);
        if (rc) return rc;
    }
    return 0;
}
