// Fused data gradient of a dense block's base channels (dgrad_block_kernels.h) in Winograd F(2x2, 3x3) form.
//
// A data gradient is a 3x3 correlation of the dY maps with the flipped filter, so the transform of the forward kernel
// (wino_fwd_kernels.h) applies unchanged: per step (one 16-channel group of the block's input, one layer of the block)
//     dX[tile][channel] = A^T [ sum_c U_xi[c][channel] .* V_xi[tile][c] ] A ,   c = the layer's 12 dY maps,
// 16 MFMAs per 4-map quad instead of 27 for the same 32 x 2 pixels: 48 instead of 108 MFMAs per 64 pixels and step.  fp32 MFMA
// and VALU cycles add up on this part, so what counts is the sum: ~1536 MFMA cycles + ~250 VALU instructions per
// step and wave here against 2592 + ~150 for 48 pixels in the direct kernel -- about 0.6x the ALU time per pixel.
//
// Structure (as dgrad_block8_kernel): a 512-thread block owns a 32 x 8 pixel tile; the NL*12 prepared dY maps of the tile (+1
// halo) are LDS-resident for the whole block; its two halves (4 waves = the 4 tile rows each) work on DIFFERENT 16-channel groups
// of every step, each half with its own double-buffered slice of transformed weights U (12 KB per (layer, group), one contiguous
// 16-byte DMA -- they are transformed once per backward pass by dgrad_wino_weights_kernel).  The input transform B^T d B of the
// lane's own 4x4 dY patch is recomputed per step (VALU, no BN on this side); the output transform, the layer's ReLU mask / BN
// backward and the accumulation over the block's layers happen per lane in registers, one read of x and one read-modify-write
// of the gradient buffer per channel as in the direct kernel.  LDS: 66 KB (dY) + 48 KB (U) + 1 KB: one block of 8 waves per CU.
#pragma once

#include "dgrad_block_kernels.h"

namespace endo {

constexpr int kWinoDgradSlice = 16 * 12 * 16;          // floats of U per (layer, 16-channel group): [xi][c][j]
constexpr int kWinoDgradMaxLayers = 48;

struct WinoDgradTable {
    int layers;
    int start[kWinoDgradMaxLayers + 1];          // prefix sum of 16 * groups * 12 work items
    int cin[kWinoDgradMaxLayers];
    int groups[kWinoDgradMaxLayers];             // 16-channel groups of the layer's input that are transformed (the block's base channels)
    int64_t w_off[kWinoDgradMaxLayers];          // floats from the parameter base: W[12][cin][3][3]
    int64_t u_off[kWinoDgradMaxLayers];          // floats from the U base
};

// U[group][xi][c][j] = (G g' G^T)[xi],  g'[a][b] = W[c][16 group + j][2 - a][2 - b]  (the flipped filter of the data gradient)
// layout 0: [group][xi][c][j] (dgrad_wino8_kernel: dword B reads);  layout 1: [group][c][a][j][i], xi = 4 a + i (dgrad_wino3_kernel:
// one 16-byte B read per transform row)
// layout1_max_groups: layers whose block has at most this many base-channel groups get layout 1 (0: layout 0 everywhere)
static __global__ void __launch_bounds__(256) dgrad_wino_weights_kernel(const WinoDgradTable t, const float* __restrict__ params, float* __restrict__ u,
                                                                 int layout1_max_groups = 0) {
    const int total = t.start[t.layers];
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += gridDim.x * blockDim.x) {
        int l = 0;
        while (item >= t.start[l + 1]) ++l;
        const int e = item - t.start[l];
        const int j = e & 15, c = (e >> 4) % 12, grp = (e >> 4) / 12;
        const int ci = grp * 16 + j;
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = 0.f;
        if (ci < t.cin[l]) {
            const float* src = params + t.w_off[l] + (static_cast<int64_t>(c) * t.cin[l] + ci) * 9;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) g[a][b] = src[(2 - a) * 3 + (2 - b)];
        }
        float h[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            h[0][b] = g[0][b];
            h[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            h[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            h[3][b] = g[2][b];
        }
        if (t.groups[l] <= layout1_max_groups) {
            float* dst = u + t.u_off[l] + static_cast<int64_t>(grp) * kWinoDgradSlice + (c * 64 + j) * 4;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                *reinterpret_cast<f32x4*>(dst + a * 64) = f32x4{h[a][0], 0.5f * (h[a][0] + h[a][1] + h[a][2]), 0.5f * (h[a][0] - h[a][1] + h[a][2]), h[a][2]};
            continue;
        }
        float* dst = u + t.u_off[l] + static_cast<int64_t>(grp) * kWinoDgradSlice + c * 16 + j;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            dst[(4 * a + 0) * 192] = h[a][0];
            dst[(4 * a + 1) * 192] = 0.5f * (h[a][0] + h[a][1] + h[a][2]);
            dst[(4 * a + 2) * 192] = 0.5f * (h[a][0] - h[a][1] + h[a][2]);
            dst[(4 * a + 3) * 192] = h[a][2];
        }
    }
}

template <int NL>
struct DgradWino8Geom {
    static constexpr int kThreads = 512;
    static constexpr int kTileX = 32, kTileY = 8;
    static constexpr int kRows = kTileY + 2, kCols = kTileX + 2;
    static constexpr int kPlane = kRows * kCols;                        // 340
    static constexpr int kCS = 352;                                     // map stride == 32 (mod 64) dwords: the 8-byte patch reads of the 4 maps of a quad hit disjoint banks
    static_assert(kCS >= kPlane && kCS % 64 == 32, "dY map stride");
    static constexpr int kU = kWinoDgradSlice;
    static constexpr int kUUnits = kU / 4;                              // 768 float4 units: 3 per thread of a half
    static constexpr int kRed = 4 * 16 * 2;                             // per (buffer, half): [4 waves][16][2]
    static constexpr size_t kBytes = sizeof(float) * (NL * 12 * kCS + 2 * 2 * kU + 2 * 2 * kRed);
    static_assert(kBytes <= 160 * 1024, "one block per CU");
};

// p.w % 32 == 0, p.h % 8 == 0, p.count % 16 == 0 (whole tiles and groups); u[l]: the layer's transformed weights, group-major.
// EXP: diagnostic bit mask for tools/wino_bench (0 in the library; timing only): 1 = no x / gradient loads, 2 = no stores, 4 = no BN-sum
// atomics, 8 = weights DMA'd for the first step only, 16 = no per-step barrier / DMA wait (racy), 32 = no MFMAs, 64 = no input
// transform, 128 = trivial epilogue (no output transform / mask / sums)
template <int NL, int EXP = 0>
__global__ void __launch_bounds__(512, 2) dgrad_wino8_kernel(const DgradBlockParams p0, const float* __restrict__ u0, const float* __restrict__ u1,
                                                             const float* __restrict__ u2, const float* __restrict__ u3) {
    using G = DgradWino8Geom<NL>;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int grp = p0.group_n > 0 ? blockIdx.z / p0.group_n : 0;
    const int n = blockIdx.z - grp * p0.group_n;
    const DgradBlockParams& p = p0;
    const int64_t grp_off = grp * p0.gs;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_g = smem;                               // [NL*12][kCS]
    float* s_u = s_g + NL * 12 * G::kCS;             // [half][2][xi 16][c 12][j 16]
    float* s_red = s_u + 4 * G::kU;                  // [2][half][4 waves][16][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, w4 = wave & 3;       // w4 = tile row of the wave
    const int th = tid & 255;
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int tile = (gridDim.x & 7) == 0 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int x0 = (tile % p.tiles_x) * G::kTileX;
    const int y0 = (tile / p.tiles_x) * G::kTileY;
    const int px = x0 + 8 * lk;                      // the lane's 8 output columns (tiles 4 lk .. 4 lk + 3)
    const int py = y0 + 2 * w4;                      // and its 2 output rows
    const int ngroups = p.count / 16;
    const int g_per = (ngroups + gridDim.y - 1) / gridDim.y;
    const int g_begin = blockIdx.y * g_per;
    const int g_end = min(ngroups, g_begin + g_per);
    if (g_begin >= g_end) return;
    const int npairs = (g_end - g_begin + 1) / 2;
    const int nsteps = npairs * NL;
    const float* const u_layer[4] = {u0, u1, u2, u3};

    // ---- dY tile: NL*12 maps with a 1-pixel halo, dword DMA by all 8 waves (once per block) ----
    {
        int goff = 0;
        bool ok = false;
        if (tid < G::kPlane) {
            const int ry = tid / G::kCols, rx = tid - ry * G::kCols;
            const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
            if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) { ok = true; goff = gy * p.g_w + gx; }
        }
        const float* g_n = p.g + grp_off + n * p.g_ns;
        const int e0 = wave * 64;
        if (e0 < G::kPlane) {
            for (int c = 0; c < NL * 12; ++c) {
                const float* src = ok ? g_n + static_cast<int64_t>(c) * p.g_cs + goff : g_pad_consts + 4;
                if (e0 + lane < G::kPlane) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_g + c * G::kCS + e0), 4, 0, 0);
            }
        }
    }

    auto step_group = [&](int step) { return g_begin + 2 * (step / NL) + half; };
    // this half's U slice of a step: one contiguous 12 KB run, 3 float4 units per thread
    auto issue_weights = [&](int step, int buf) {
        const int gq = step_group(step), l = step % NL;
        if (gq >= g_end) return;                                     // half-uniform
        const float* src = u_layer[l] + static_cast<int64_t>(gq) * G::kU + 4 * th;
        float* dst = s_u + (half * 2 + buf) * G::kU + w4 * 256;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + k * 1024), (lptr_t)(dst + k * 1024), 16, 0, 0);
    };

    const float* x_n = p.x + grp_off + n * p.ns;
    float* out_n = p.out + grp_off + n * p.ns;
    f32x4 xc[2][2], dc[2][2], total[2][2];           // [row][column half]: 8 consecutive pixels of 2 rows
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) total[r][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue_weights(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const int gq = step_group(step), l = step % NL;
        const bool active = gq < g_end;                              // half-uniform
        const bool last_layer = (l == NL - 1);
        const int co = gq * 16 + li;
        float scale = 0.f, beta = 0.f, mean = 0.f, rstd = 0.f;
        if (active) {
            if (l == 0) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int64_t o = static_cast<int64_t>(co) * p.cs + static_cast<int64_t>(py + r) * p.w + px + 4 * hh;
                        if constexpr ((EXP & 1) != 0) {
                            xc[r][hh] = f32x4{0.1f * lane, 0.2f, -0.3f, 0.4f}; dc[r][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
                        } else {
                            xc[r][hh] = *reinterpret_cast<const f32x4*>(x_n + o);
                            dc[r][hh] = co >= p.acc_from ? *reinterpret_cast<const f32x4*>(out_n + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
            }
            mean = p.saved[l][grp_off + 2 * co]; rstd = p.saved[l][grp_off + 2 * co + 1];
            scale = p.gamma[l][co] * rstd;
            beta = p.beta[l][co];
        }
        if (step + 1 < nsteps && (!(EXP & 8) || step == 0)) issue_weights(step + 1, buf ^ 1);
        if (active) {
            // ---- 16 transform-domain GEMMs over the layer's 12 dY maps: M = this wave's row of 16 tiles, N = the half's 16 channels ----
            f32x4 acc[16];
            const float* ub = s_u + (half * 2 + buf) * G::kU;
#pragma unroll
            for (int quad = 0; quad < 3; ++quad) {
                // the lane's 4x4 patch of map (l, 4 quad + lk): LDS rows 2 w4 .. 2 w4 + 3, columns 2 li .. 2 li + 3 (two aligned pairs)
                const float* a_base = s_g + (l * 12 + quad * 4 + lk) * G::kCS + (2 * w4) * G::kCols + 2 * li;
                const float* b_base = ub + (quad * 4 + lk) * 16 + li;
                f32x2 lo[4], hi[4];
#pragma unroll
                for (int row = 0; row < 4; ++row) {
                    lo[row] = *reinterpret_cast<const f32x2*>(a_base + row * G::kCols);
                    hi[row] = *reinterpret_cast<const f32x2*>(a_base + row * G::kCols + 2);
                }
                f32x2 tl[4], th2[4];          // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
                tl[0] = lo[0] - lo[2]; th2[0] = hi[0] - hi[2];
                tl[1] = lo[1] + lo[2]; th2[1] = hi[1] + hi[2];
                tl[2] = lo[2] - lo[1]; th2[2] = hi[2] - hi[1];
                tl[3] = lo[1] - lo[3]; th2[3] = hi[1] - hi[3];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float c0 = tl[a][0], c1 = tl[a][1], c2 = th2[a][0], c3 = th2[a][1];
                    float v0 = c0 - c2, v1 = c1 + c2, v2 = c2 - c1, v3 = c1 - c3;
                    if constexpr ((EXP & 64) != 0) { v0 = lo[a][0]; v1 = lo[a][1]; v2 = hi[a][0]; v3 = hi[a][1]; }
                    const float b0 = b_base[(4 * a + 0) * 192], b1 = b_base[(4 * a + 1) * 192];
                    const float b2 = b_base[(4 * a + 2) * 192], b3 = b_base[(4 * a + 3) * 192];
                    if constexpr ((EXP & 32) != 0) {
                        if (quad == 0) {
                            acc[4 * a + 0] = f32x4{v0 * b0, 0.f, 0.f, 0.f}; acc[4 * a + 1] = f32x4{v1 * b1, 0.f, 0.f, 0.f};
                            acc[4 * a + 2] = f32x4{v2 * b2, 0.f, 0.f, 0.f}; acc[4 * a + 3] = f32x4{v3 * b3, 0.f, 0.f, 0.f};
                        } else {
                            acc[4 * a + 0][quad] += v0 * b0; acc[4 * a + 1][quad] += v1 * b1;
                            acc[4 * a + 2][quad] += v2 * b2; acc[4 * a + 3][quad] += v3 * b3;
                        }
                    } else
                    if (quad == 0) {
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        acc[4 * a + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, b0, zero, 0, 0, 0);
                        acc[4 * a + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, b1, zero, 0, 0, 0);
                        acc[4 * a + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v2, b2, zero, 0, 0, 0);
                        acc[4 * a + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v3, b3, zero, 0, 0, 0);
                    } else {
                        acc[4 * a + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, b0, acc[4 * a + 0], 0, 0, 0);
                        acc[4 * a + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, b1, acc[4 * a + 1], 0, 0, 0);
                        acc[4 * a + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v2, b2, acc[4 * a + 2], 0, 0, 0);
                        acc[4 * a + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v3, b3, acc[4 * a + 3], 0, 0, 0);
                    }
                }
            }
            // ---- output transform A^T M A (tiles 4 lk + e), then layer l's ReLU mask + BN backward, accumulated over the layers ----
            float s1 = 0.f, s2 = 0.f;
            if constexpr ((EXP & 128) != 0) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            total[r][hh][k] += acc[4 * (2 * r + hh) + k][0] + acc[4 * (2 * r + hh) + k][1] + acc[4 * (2 * r + hh) + k][2] + acc[4 * (2 * r + hh) + k][3] + xc[r][hh][k] * scale;
                s1 = total[0][0][0]; s2 = mean + beta;
            } else
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u0r[4], u1r[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float m0 = acc[c][e], m1 = acc[4 + c][e], m2 = acc[8 + c][e], m3 = acc[12 + c][e];
                    u0r[c] = m0 + m1 + m2;
                    u1r[c] = m1 - m2 - m3;
                }
                const float d[2][2] = {{u0r[0] + u0r[1] + u0r[2], u0r[1] - u0r[2] - u0r[3]},
                                       {u1r[0] + u1r[1] + u1r[2], u1r[1] - u1r[2] - u1r[3]}};
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int cx = 0; cx < 2; ++cx) {
                        const int hh = e >> 1, k = 2 * (e & 1) + cx;
                        const float xcen = xc[r][hh][k] - mean;
                        const float z = fmaf(xcen, scale, beta);
                        const float dz = z > 0.f ? d[r][cx] : 0.f;
                        s1 += dz;
                        s2 = fmaf(dz, xcen, s2);
                        total[r][hh][k] = fmaf(dz, scale, total[r][hh][k]);
                    }
            }
            if (last_layer) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        f32x4 o = dc[r][hh];
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] += total[r][hh][k];
                        if constexpr ((EXP & 2) != 0) asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
                        else
                        *reinterpret_cast<f32x4*>(out_n + static_cast<int64_t>(co) * p.cs + static_cast<int64_t>(py + r) * p.w + px + 4 * hh) = o;
                        total[r][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
            }
            s2 *= rstd;
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lk == 0) {
                float* red = s_red + (buf * 2 + half) * G::kRed;
                red[(w4 * 16 + li) * 2] = s1;
                red[(w4 * 16 + li) * 2 + 1] = s2;
            }
        }
        if (!(EXP & 16) || step + 1 >= nsteps) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (!(EXP & 4) && active && th < 32) {
            const int j = th >> 1, which = th & 1;
            const int cj = gq * 16 + j;
            const float* red = s_red + (buf * 2 + half) * G::kRed;
            double t = 0.0;
            for (int wv = 0; wv < 4; ++wv) t += static_cast<double>(red[(wv * 16 + j) * 2 + which]);
            atomicAdd(p.scratch[l] + bn_slot_offset(p.slot_stride) + grp_off / 2 + 2 * cj + which, t);
        }
    }
}

inline bool dgrad_wino_ok(const DgradBlockParams& p) {
    return dgrad_block_ok(p) && (p.w % 32 == 0) && (p.h % 8 == 0) && (p.count % 16 == 0) && p.count >= 32;
}

// u[l]: transformed weights of layer l of the block (group-major slices of kWinoDgradSlice floats)
template <int NL, int EXP = 0>
inline int launch_dgrad_wino8(DgradBlockParams p, const float* const (&u)[4], hipStream_t stream) {
    using G = DgradWino8Geom<NL>;
    p.tiles_x = p.w / G::kTileX;
    const int tiles_y = p.h / G::kTileY;
    static bool configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& configured = configured_by_device[dev & 15];
    if (!configured) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_wino8_kernel<NL, EXP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(G::kBytes)));
        configured = true;
    }
    dgrad_wino8_kernel<NL, EXP><<<dim3(p.tiles_x * tiles_y, 1, p.n), G::kThreads, G::kBytes, stream>>>(p, u[0], u[1], u[2], u[3]);
    ENDO_LAUNCH_CHECK();
    return 0;
}

}  // namespace endo
