// Dense-layer forward (BN -> ReLU -> conv3x3, growth 12; reference models.py:19-28) in Winograd F(4x4, 3x3) form on the fp32 matrix
// cores: 36 instead of 144 multiply-accumulates per 4x4 output tile, input channel and output channel (F(2x2, 3x3), wino_fwd_kernels.h: 64).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A          d_c: 6x6 patch of relu(bn(x_c)), g_c: 3x3 filter, Y: 4x4 outputs
//   Interpolation points 0, +-5/8, +-3/2, inf (round 5; the weight gradient keeps the textbook 0, +-1, +-2, inf): what separates an fp32
//   F(4x4, 3x3) from the direct form is the fp32 sum over the input channels of transform-domain products whose magnitudes the input transform
//   has inflated and the output transform cancels again (DESIGN_HISTORY.md 4.19), and that inflation depends on the points.  Emulated in fp32
//   against fp64 (one layer, 48..180 channels of post-ReLU inputs, Kaiming weights): rms 0.5-1.1e-6 / max 1.3-2.3e-6 of the output's maximum
//   with these points, 1.5-2.4e-6 / 4.4-9.5e-6 with +-1, +-2 (the direct fp32 sum: max 1.7e-6).  All constants are dyadic rationals (exact in
//   fp32) except the three row scales of G; same instruction count in the K loop.
//
// 36 independent GEMMs over the input channels, one per transform-domain position xi (v_mfma_f32_16x16x4_f32):
//     A[i = tile][k = channel] = V_xi = (B^T d B)[xi]   computed by the lane that owns (tile i, channel k) from ITS 6x6 patch
//     B[k = channel][j = cout] = U_xi = (G g G^T)[xi]   pre-transformed once per forward pass (wino4_fwd_weights_kernel)
//     D_xi[i][j] accumulates over the whole K loop (144 registers); the output transform A^T M A is per lane, in registers.
// A wave owns a row of 16 tiles (64 x 4 output pixels), a block four such rows (64 x 16).  K-chunks of 4 input channels (the haloed
// 72 x 18 tile of each + the chunk's U slice, 30 KB) arrive by buffer_load ... lds into two LDS stages, one barrier per chunk; pixels
// outside the image are out of the descriptor's range (zeros) and are zeroed again after BN + ReLU (row by row where a whole patch row
// is outside, through the BN constants for the halo columns).  Per-channel sum / sum^2 of the stored values for the BN layers that follow.
//
// Rounding: the transforms carry factors up to 8 (A^T) / 5 (B^T) / 1/24 (G); against fp64 the network's output is within 6e-6 of its maximum
// with every dense layer in this form (4e-6 .. 6e-6 at 64x96 .. 256x320; direct and F(2x2, 3x3): 8e-7 .. 1e-6) -- inside the 1e-4 the
// parity target states (the default form of level 0 since round 5; ENDO_OPT_WINO_FWD = 1 keeps F(2x2, 3x3) selectable).
#pragma once

#include "conv_dma_kernels.h"
#include "wgrad_f34_kernels.h"
#include "wino_fwd_kernels.h"

namespace endo {

// interpolation points +-a, +-b of the forward (0 and infinity besides) and what the three transforms need of them
constexpr float kW4A = 0.625f, kW4B = 1.5f;
constexpr float kW4A2 = kW4A * kW4A, kW4B2 = kW4B * kW4B;                 // 25/64, 9/4
constexpr float kW4A3 = kW4A2 * kW4A, kW4B3 = kW4B2 * kW4B;               // 125/512, 27/8
constexpr float kW4P = kW4A2 * kW4B2, kW4S = kW4A2 + kW4B2;               // 225/256, 169/64: B^T rows 0 and 5 are (P, 0, -S, 0, 1, 0) / (0, P, 0, -S, 0, 1)
constexpr float kW4N0 = 1.f / kW4P;                                        // 1 / N_0
constexpr float kW4NA = 1.f / (2.f * kW4A2 * (kW4A2 - kW4B2));             // 1 / N_a = 1 / N_-a
constexpr float kW4NB = 1.f / (2.f * kW4B2 * (kW4B2 - kW4A2));             // 1 / N_b = 1 / N_-b

constexpr int kW4UStride = 576;          // floats per input channel of U: [row pair 3][j 16][2 x 6]: a lane's two rows are three 16-byte reads

// ---- weights: U[ci][i >> 1][j][6 (i & 1) + jj] = (G g G^T)[i][jj], g = W[j][ci][3][3]; columns j >= cout are zero --------------------
__global__ void __launch_bounds__(256) wino4_fwd_weights_kernel(const WinoWeightTable t, const float* __restrict__ params, float* __restrict__ u) {
    const int total = t.start[t.layers];
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += gridDim.x * blockDim.x) {
        int l = 0;
        while (item >= t.start[l + 1]) ++l;
        const int e = item - t.start[l];
        const int ci = e >> 4, j = e & 15;
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = 0.f;
        if (j < t.cout[l]) {
            const float* src = params + t.w_off[l] + (static_cast<int64_t>(j) * t.cin[l] + ci) * 9;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) g[a][b] = src[a * 3 + b];
        }
        // G: row j = (1, p_j, p_j^2) / N_j, N_j = prod over the other finite points of (p_j - p_l); (0, 0, 1) for the point at infinity
        auto gt = [](float a, float b, float c, float (&o)[6]) {
            const float ea = fmaf(kW4A2, c, a), eb = fmaf(kW4B2, c, a);          // a + p^2 c
            o[0] = kW4N0 * a;
            o[1] = kW4NA * fmaf(kW4A, b, ea);
            o[2] = kW4NA * fmaf(-kW4A, b, ea);
            o[3] = kW4NB * fmaf(kW4B, b, eb);
            o[4] = kW4NB * fmaf(-kW4B, b, eb);
            o[5] = c;
        };
        float h[6][3];          // G g
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float o[6];
            gt(g[0][b], g[1][b], g[2][b], o);
#pragma unroll
            for (int i = 0; i < 6; ++i) h[i][b] = o[i];
        }
        float* dst = u + t.u_off[l] + static_cast<int64_t>(ci) * kW4UStride + j * 12;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float o[6];
            gt(h[i][0], h[i][1], h[i][2], o);
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) dst[(i >> 1) * 192 + (i & 1) * 6 + jj] = o[jj];
        }
    }
}

struct Wino4Geom {
    static constexpr int kTileX = 64, kTileY = 16;
    static constexpr int kCols = kTileX + 8;                  // the tile starts 4 pixels left of the output tile: rows of whole aligned float4s
    static constexpr int kRows = kTileY + 2;
    static constexpr int kPlane = kRows * kCols;              // 1296 floats = 324 DMA units per channel
    static constexpr int kKC = 4;
    static constexpr int kXUnits = kKC * kPlane / 4;          // 1296
    static constexpr int kUUnits = kKC * kW4UStride / 4;      // 576
    static constexpr int kXRounds = (kXUnits + kConvThreads - 1) / kConvThreads;          // 6 (the last one: 16 units)
    static constexpr int kURounds = (kUUnits + kConvThreads - 1) / kConvThreads;          // 3 (the last one: 64 units)
    static constexpr int kBuf = kKC * kPlane + kKC * kW4UStride;          // floats per stage
    static constexpr int kTail = 4 * 16 * 2;                  // statistics scratch: [4 waves][16][2]
    static size_t bytes(int bn_cap) { return sizeof(float) * (2 * kBuf + 4 * bn_cap + kTail); }          // 3 BN tables + the final-conv weights of FIN
};

// 6 -> 4 output transform A^T m: rows (1 1 1 1 1 0), (0 a -a b -b 0), (0 a^2 a^2 b^2 b^2 0), (0 a^3 -a^3 b^3 -b^3 1)
__device__ __forceinline__ void w4_at(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5, float (&y)[4]) {
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y[0] = m0 + s1 + s2;
    y[1] = fmaf(kW4B, d2, kW4A * d1);
    y[2] = fmaf(kW4B2, s2, kW4A2 * s1);
    y[3] = fmaf(kW4B3, d2, fmaf(kW4A3, d1, m5));
}

// 6-point input transform B^T d on a value or a packed pair: rows (P 0 -S 0 1 0), (0 -ab^2 -b^2 a 1 0), (0 ab^2 -b^2 -a 1 0),
// (0 -a^2b -a^2 b 1 0), (0 a^2b -a^2 -b 1 0), (0 P 0 -S 0 1) -- 12 fused multiply-adds, as f34_bt for the points +-1, +-2
template <typename T>
__device__ __forceinline__ void w4_bt(const T d0, const T d1, const T d2, const T d3, const T d4, const T d5, T (&t)[6]) {
    t[0] = f34_fma<T>(kW4P, d0, f34_fma<T>(-kW4S, d2, d4));
    const T pp = f34_fma<T>(-kW4B2, d2, d4), qh = f34_fma<T>(-kW4B2, d1, d3);
    t[1] = f34_fma<T>(kW4A, qh, pp);
    t[2] = f34_fma<T>(-kW4A, qh, pp);
    const T rr = f34_fma<T>(-kW4A2, d2, d4), sh = f34_fma<T>(-kW4A2, d1, d3);
    t[3] = f34_fma<T>(kW4B, sh, rr);
    t[4] = f34_fma<T>(-kW4B, sh, rr);
    t[5] = f34_fma<T>(kW4P, d1, f34_fma<T>(-kW4S, d3, d5));
}

// p.wgt = this layer's U (kW4UStride floats per input channel), p.cout <= 16, p.w % 4 == 0, p.h % 4 == 0, p.cin % 4 == 0, 16-byte aligned planes
// FIN: the launch is the network's LAST dense layer and also forms the final 1x1 convolution's sum over its input channels (ConvParams::fin_w,
// fin_out): every raw input value passes through a lane's patch exactly once as an interior pixel of its tile, so the 180 of the final
// convolution's 192 planes this kernel streams anyway are not read a second time (final_fwd_kernel then adds the 12 new maps and the bias).
template <bool FIN = false>
__global__ void __launch_bounds__(kConvThreads, 2) wino4_fwd_kernel(const ConvParams p0) {
    using G = Wino4Geom;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_aux = smem + 2 * G::kBuf;
    int grp, n;
    group_of(p0, blockIdx.z, grp, n);
    const ConvParams p = group_view(p0, grp);
    const int groups = p0.group_n > 0 ? gridDim.z / p0.group_n : 1;
    const bool first_of_group = blockIdx.x == 0 && n == 0;
    const int cap = p.bn_cap;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int tile = (gridDim.x & 7) == 0 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int x0 = (tile % p.tiles_x) * G::kTileX;
    const int y0 = (tile / p.tiles_x) * G::kTileY;

    for (int c = tid; c < p.cin; c += kConvThreads) {
        float scale, mean, beta;
        bn_input_constants(p, p0, grp, groups, first_of_group, c, scale, mean, beta);
        s_aux[c] = scale;
        s_aux[cap + c] = mean;
        s_aux[2 * cap + c] = beta;
        if constexpr (FIN) s_aux[3 * cap + G::kTail + c] = p.fin_w[c];
    }
    f32x4 fin[4];          // FIN: sum over this lane's channels (4 chunk + lk) of w_final[c] * x_raw[c] at its tile's 4 x 4 pixels
#pragma unroll
    for (int r = 0; r < 4; ++r) fin[r] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 acc[36];
#pragma unroll
    for (int xi = 0; xi < 36; ++xi) acc[xi] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this thread's DMA units of a chunk (the same every chunk): byte offsets inside the sample / the layer's U, advanced per chunk by a
    // scalar; a unit outside the image points past the descriptor's range (zeros)
    const __amdgpu_buffer_rsrc_t xr = f34_rsrc(p.in + n * p.in_ns, p.cin * p.in_cs * 4);
    const __amdgpu_buffer_rsrc_t ur = f34_rsrc(p.wgt, p.cin * kW4UStride * 4);
    unsigned x_vo[G::kXRounds];
#pragma unroll
    for (int k = 0; k < G::kXRounds; ++k) {
        const int e = tid + k * kConvThreads;
        x_vo[k] = 0x80000000u;
        if (e < G::kXUnits) {
            const int c = e / (G::kPlane / 4), u = e - c * (G::kPlane / 4);
            const int ry = u / (G::kCols / 4);
            const int rx = (u - ry * (G::kCols / 4)) * 4;
            const int gy = y0 - 1 + ry;
            const int gx = x0 - 4 + rx;
            if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) x_vo[k] = 4u * static_cast<unsigned>(c * p.in_cs + gy * p.in_w + gx);
        }
    }
    const unsigned u_vo = 16u * static_cast<unsigned>(tid);
    const int nchunks = p.cin / G::kKC;

    auto issue_dma = [&](int chunk, int buf) {
        float* s_stage = smem + buf * G::kBuf + wave * 256;          // round k: this wave's 64 units land at 16 (256 k + 64 wave) bytes and up
        const unsigned xs = 4u * static_cast<unsigned>(chunk * G::kKC * p.in_cs);
        const unsigned us = 4u * static_cast<unsigned>(chunk * G::kKC * kW4UStride);
#pragma unroll
        for (int k = 0; k < G::kXRounds; ++k) {
            const int e0 = k * kConvThreads + wave * 64;
            if (e0 < G::kXUnits) {          // (wave-uniform)
                if (e0 + 64 <= G::kXUnits || e0 + lane < G::kXUnits)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr_t)(s_stage + k * 4 * kConvThreads), 16, x_vo[k], xs, 0, 0);
            }
        }
        float* s_u = s_stage + G::kKC * G::kPlane;
#pragma unroll
        for (int k = 0; k < G::kURounds; ++k) {
            const int e0 = k * kConvThreads + wave * 64;
            if (e0 < G::kUUnits)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (lptr_t)(s_u + k * 4 * kConvThreads), 16, u_vo + 16u * static_cast<unsigned>(k * kConvThreads), us, 0, 0);
        }
    };

    // borders of this lane's patch: the halo column left of tile 0 at the image's left edge, right of the last tile inside the image;
    // patch rows above / below the image belong to the first / last tile row (wave-uniform)
    const bool l_out = x0 + 4 * li == 0, r_out = x0 + 4 * li + 4 >= p.w;
    const int py = y0 + 4 * wave - 1;          // image row of patch row 0
    const bool top_out = py < 0, bottom_out = py + 5 >= p.h;

    auto compute = [&](int chunk, int buf) {
        const float* s_in = smem + buf * G::kBuf;
        const float* s_u = s_in + G::kKC * G::kPlane;
        const int ch = chunk * G::kKC + lk;
        const float sc = s_aux[ch], mn = s_aux[cap + ch], bt = s_aux[2 * cap + ch];
        const float sh = fmaf(-mn, sc, bt);
        const f32x2 sc_m = {sc, sc}, sh_m = {sh, sh};
        const f32x2 sc_e = {l_out ? 0.f : sc, r_out ? 0.f : sc}, sh_e = {l_out ? 0.f : sh, r_out ? 0.f : sh};
        const float* a_base = s_in + lk * G::kPlane + (4 * wave) * G::kCols + 4 * li + 4;
        float wfin = 0.f;
        if constexpr (FIN) wfin = s_aux[3 * cap + G::kTail + ch];
        float d[6][6];          // [row][column in the order 0, 5, 1, 2, 3, 4]
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const f32x4 m = *reinterpret_cast<const f32x4*>(a_base + r * G::kCols);
            if constexpr (FIN) {
                if (r >= 1 && r <= 4) {          // patch rows 1..4, columns 1..4 = the tile's own pixels (raw values; zeros outside the image)
#pragma unroll
                    for (int k = 0; k < 4; ++k) fin[r - 1][k] = fmaf(wfin, m[k], fin[r - 1][k]);
                }
            }
            const float hl = a_base[r * G::kCols - 1], hr = a_base[r * G::kCols + 4];
            const f32x2 e = __builtin_elementwise_fma(f32x2{hl, hr}, sc_e, sh_e);
            const f32x2 a = __builtin_elementwise_fma(f32x2{m[0], m[1]}, sc_m, sh_m);
            const f32x2 b = __builtin_elementwise_fma(f32x2{m[2], m[3]}, sc_m, sh_m);
            d[r][0] = fmaxf(e[0], 0.f); d[r][1] = fmaxf(e[1], 0.f);
            d[r][2] = fmaxf(a[0], 0.f); d[r][3] = fmaxf(a[1], 0.f);
            d[r][4] = fmaxf(b[0], 0.f); d[r][5] = fmaxf(b[1], 0.f);
        }
        if (top_out) {
#pragma unroll
            for (int e = 0; e < 6; ++e) d[0][e] = 0.f;
        }
        if (bottom_out) {
#pragma unroll
            for (int e = 0; e < 6; ++e) d[5][e] = 0.f;
        }
        // column pass on the column pairs (0, 5), (1, 2), (3, 4)
        f32x2 t05[6], t12[6], t34[6];
        w4_bt(f32x2{d[0][0], d[0][1]}, f32x2{d[1][0], d[1][1]}, f32x2{d[2][0], d[2][1]}, f32x2{d[3][0], d[3][1]}, f32x2{d[4][0], d[4][1]},
              f32x2{d[5][0], d[5][1]}, t05);
        w4_bt(f32x2{d[0][2], d[0][3]}, f32x2{d[1][2], d[1][3]}, f32x2{d[2][2], d[2][3]}, f32x2{d[3][2], d[3][3]}, f32x2{d[4][2], d[4][3]},
              f32x2{d[5][2], d[5][3]}, t12);
        w4_bt(f32x2{d[0][4], d[0][5]}, f32x2{d[1][4], d[1][5]}, f32x2{d[2][4], d[2][5]}, f32x2{d[3][4], d[3][5]}, f32x2{d[4][4], d[4][5]},
              f32x2{d[5][4], d[5][5]}, t34);
        const float* b_base = s_u + lk * kW4UStride + li * 12;
#pragma unroll
        for (int ip = 0; ip < 3; ++ip) {
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(b_base + ip * 192);
            const f32x4 u1 = *reinterpret_cast<const f32x4*>(b_base + ip * 192 + 4);
            const f32x4 u2 = *reinterpret_cast<const f32x4*>(b_base + ip * 192 + 8);
            const float uu[12] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3], u2[0], u2[1], u2[2], u2[3]};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * ip + h;
                const f32x2 p05 = t05[i], p12 = t12[i], p34 = t34[i];
                // the same transform along the row (columns in the order 0, 5 | 1, 2 | 3, 4)
                const float x0v = fmaf(kW4P, p05[0], fmaf(-kW4S, p12[1], p34[1]));
                const float x5v = fmaf(kW4P, p12[0], fmaf(-kW4S, p34[0], p05[1]));
                const f32x2 qp = f34_fma<f32x2>(-kW4B2, p12, p34);        // (T3 - b^2 T1, T4 - b^2 T2)
                const f32x2 sr = f34_fma<f32x2>(-kW4A2, p12, p34);        // (T3 - a^2 T1, T4 - a^2 T2)
                const float x1v = fmaf(kW4A, qp[0], qp[1]), x2v = fmaf(-kW4A, qp[0], qp[1]);
                const float x3v = fmaf(kW4B, sr[0], sr[1]), x4v = fmaf(-kW4B, sr[0], sr[1]);
                acc[6 * i + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0v, uu[6 * h + 0], acc[6 * i + 0], 0, 0, 0);
                acc[6 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1v, uu[6 * h + 1], acc[6 * i + 1], 0, 0, 0);
                acc[6 * i + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x2v, uu[6 * h + 2], acc[6 * i + 2], 0, 0, 0);
                acc[6 * i + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x3v, uu[6 * h + 3], acc[6 * i + 3], 0, 0, 0);
                acc[6 * i + 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(x4v, uu[6 * h + 4], acc[6 * i + 4], 0, 0, 0);
                acc[6 * i + 5] = __builtin_amdgcn_mfma_f32_16x16x4f32(x5v, uu[6 * h + 5], acc[6 * i + 5], 0, 0, 0);
            }
        }
    };

    if (nchunks > 0) issue_dma(0, 0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int b = chunk & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();           // everybody's chunk has landed; the other stage is free again
        if (chunk + 1 < nchunks) issue_dma(chunk + 1, b ^ 1);
        compute(chunk, b);
    }

    if constexpr (FIN) {
        // the four channel lanes of a tile (lk) add up; lane lk then stores row lk of the tile's 4 x 4 pixels
        f32x4 mine = fin[0];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v = fin[r][k];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (r == lk) mine[k] = v;
            }
        }
        const int fx = x0 + 4 * li, fy = y0 + 4 * wave + lk;
        if (fx < p.w && fy < p.h)
            *reinterpret_cast<f32x4*>(p.fin_out + static_cast<int64_t>(grp) * p0.gs + static_cast<int64_t>(n) * p.h * p.w + static_cast<int64_t>(fy) * p.w + fx) = mine;
    }

    // ---- output transform A^T M A per lane: tiles 4 lk + e (e = 0..3) of the wave's tile row, output channel li ----
    float* s_red = s_aux + 3 * cap;
    const int co = li;
    const bool co_ok = co < p.cout;
    const float bias = (co_ok && p.bias) ? p.bias[co] : 0.f;
    const int y = y0 + 4 * wave;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float c4[4][6];          // A^T M: [output row][column xi]
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            float o[4];
            w4_at(acc[c][e], acc[6 + c][e], acc[12 + c][e], acc[18 + c][e], acc[24 + c][e], acc[30 + c][e], o);
#pragma unroll
            for (int a = 0; a < 4; ++a) c4[a][c] = o[a];
        }
        const int px = x0 + 4 * (4 * lk + e);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float o[4];
            w4_at(c4[a][0], c4[a][1], c4[a][2], c4[a][3], c4[a][4], c4[a][5], o);
            const f32x4 v = {o[0] + bias, o[1] + bias, o[2] + bias, o[3] + bias};
            if (co_ok && px < p.w && y + a < p.h) {
                *reinterpret_cast<f32x4*>(p.out + n * p.out_ns + static_cast<int64_t>(co) * p.out_cs + static_cast<int64_t>(y + a) * p.out_w + px) = v;
#pragma unroll
                for (int k = 0; k < 4; ++k) { s1 += v[k]; s2 += v[k] * v[k]; }
            }
        }
    }
    if (p.out_sums) {
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        if (lk == 0) {
            s_red[(wave * 16 + li) * 2] = s1;
            s_red[(wave * 16 + li) * 2 + 1] = s2;
        }
        __syncthreads();
        if (tid < 32) {
            const int j = tid >> 1, which = tid & 1;
            if (j < p.cout) {
                double t = 0.0;
                for (int wv = 0; wv < 4; ++wv) t += static_cast<double>(s_red[(wv * 16 + j) * 2 + which]);
                atomicAdd(p.out_sums + 2 * j + which, t);
            }
        }
    }
}

inline bool wino4_fwd_ok(const ConvParams& p) {
    return wino_fwd_ok(p) && (p.h % 4 == 0) && static_cast<int64_t>(p.cin) * p.in_cs * 4 < (1ll << 31);
}

template <bool FIN>
inline int launch_wino4_fwd_t(ConvParams p, hipStream_t stream) {
    using G = Wino4Geom;
    p.tiles_x = (p.w + G::kTileX - 1) / G::kTileX;
    p.bn_cap = (p.cin + 15) / 16 * 16;
    const int tiles_y = (p.h + G::kTileY - 1) / G::kTileY;
    const size_t smem = G::bytes(p.bn_cap);
    static size_t configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& configured = configured_by_device[dev & 15];
    if (smem > configured) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_fwd_kernel<FIN>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
        configured = smem;
    }
    wino4_fwd_kernel<FIN><<<dim3(p.tiles_x * tiles_y, 1, p.n), kConvThreads, smem, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}
// with p.fin_w / p.fin_out set: the launch also forms the final convolution's sum over its input channels (wino4_fwd_kernel<true>)
inline int launch_wino4_fwd(const ConvParams& p, hipStream_t stream) {
    return p.fin_w ? launch_wino4_fwd_t<true>(p, stream) : launch_wino4_fwd_t<false>(p, stream);
}

}  // namespace endo
