// 3x3 weight gradient with the taps folded into the MFMA M dimension.
//
//   dW[co][ci][ky][kx] = sum_p a[ci][p] * dY[co][p - (ky-1, kx-1)]          (a = relu(bn(x)), zero outside)
//
// i.e. shift the SMALL operand (dY, 12 maps) instead of the big one.  GEMM view: M = (co, tap) =
// 108 rows for a growth-12 layer = 7 MFMA row groups (96 % full, against 75 % when M = co alone and
// the 9 taps are 9 separate GEMMs), N = 16 input channels, K = pixels.  Consequences:
//   * 7 MFMAs per pixel quad instead of 9
//   * the activation operand is read from LDS ONCE per pixel quad (BN+ReLU applied on that read),
//     not once per tap, and its tile needs no halo; the dY tile (12 maps + halo) is gathered with a
//     per-lane (co, tap) offset table
//   * both tiles arrive by 16-byte LDS-DMA into a double buffer, one barrier per tile
// A block owns a 16-channel input slice and a strided set of 32 x 8 pixel tiles; its 8 waves take one tile row
// each, are reduced through LDS at the end, and issue one fp32 atomic per dW element.
#pragma once

#include "conv_dma_kernels.h"
#include "wgrad_kernels.h"

namespace endo {

template <int COUT>
struct WgradTapsGeom {
    static constexpr int kTX = 32, kTY = 8;
    static constexpr int kM = COUT * 9;
    static constexpr int kMG = (kM + 15) / 16;               // MFMA row groups (7 for COUT = 12, 27 for 48)
    static constexpr int kInPlane = kTX * kTY;                // 256, no halo
    static constexpr int kCS = kInPlane + 4;                  // 260: 16-byte aligned rows for the DMA
    static constexpr int kDyCols = kTX + 8;                   // 4-pixel aligned halo on both sides
    static constexpr int kDyRows = kTY + 2;
    static constexpr int kDyPlane = kDyCols * kDyRows;        // 400
    static constexpr int kDS = kDyPlane + 20;                 // 420 == 4 (mod 32)
    static constexpr int kZero = 320;                         // zero rows read by the unused M rows of the last group
    static constexpr int kBuf = 16 * kCS + COUT * kDS + kZero;   // floats per buffer
    static constexpr int kInUnits = kInPlane / 4;             // 64 float4 per channel
    static constexpr int kDyUnits = kDyPlane / 4;             // 100 float4 per map
    static constexpr int kWaves = 8;                          // 512-thread blocks: LDS allows two per CU, so 4 waves per SIMD
    static constexpr size_t kRed = kWaves * kMG * 256;        // cross-wave reduction scratch (floats)
    static constexpr size_t kFloats = (2 * kBuf > static_cast<int>(kRed) ? 2 * kBuf : kRed) + 64 * kMaxGroups;
    static constexpr size_t kBytes = sizeof(float) * kFloats;
    static_assert(kBytes <= 160 * 1024, "two tile buffers must fit the 160 KiB LDS");
};

// BF: 1 = bf16 MFMA operands (ENDO_OPT_MFMA_BF16): four consecutive k-steps of a row (pixels 4 x4 + lk) in one v_mfma_f32_16x16x16_bf16
template <int COUT, int IN, int BF = 0>
__global__ void __launch_bounds__(512, 4) wgrad_taps_kernel(const WgradParams p) {
    using G = WgradTapsGeom<COUT>;
    static_assert(IN == IN_BNRELU || IN == IN_PLAIN || IN == IN_UPSAMPLE, "supported activation load paths");
    constexpr int MG = G::kMG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_cst = smem + G::kFloats - 64 * kMaxGroups;       // per group: scale, mean, beta of the block's 16 channels

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;
    // Blocks that work on the same tiles (all input-channel slices of one tile group) re-read the same dY
    // tiles: keep them on one XCD so the re-reads hit its L2.  Flattened id b = x + gridDim.x * y runs on
    // XCD b % 8; slices of tile group (b % 8 + 8 * k) are laid out along b / 8.
    int slice = blockIdx.x, group = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = b & 7, idx = b >> 3;
        slice = idx % gridDim.x;
        group = xcd + 8 * (idx / gridDim.x);
    }
    const int ci_base = slice * 16;
    const int co_base = blockIdx.z * COUT;          // wider convs (48 outputs) run as several COUT-wide slices
    const int tiles_per_sample = p.tiles_x * p.tiles_y;
    const int tiles_total = tiles_per_sample * p.n;

    if (tid < 16 * wg_groups(p)) {
        const int g = tid >> 4, t = tid & 15;
        const int c = ci_base + t;
        float scale = 1.f, mean = 0.f, beta = 0.f;
        if (IN == IN_BNRELU && c < p.cin) {
            const float* saved = p.saved + g * p.gs;
            mean = saved[2 * c];
            scale = p.gamma[c] * saved[2 * c + 1];
            beta = p.beta[c];
        }
        s_cst[64 * g + t] = scale; s_cst[64 * g + 16 + t] = mean; s_cst[64 * g + 32 + t] = beta;
    }

    // per-lane gather offsets into the dY tile: row m = 16 g + li = co * 9 + ky * 3 + kx reads
    // dY[co][y + 1 - ky][x + 1 - kx]  ->  tile row (y + 2 - ky), tile col (x + 5 - kx)  (halo 1 row, 4 cols)
    int aoff[MG];
#pragma unroll
    for (int g = 0; g < MG; ++g) {
        const int m = 16 * g + li;
        const int co = m / 9, tap = m - co * 9;
        const int ky = tap / 3, kx = tap - ky * 3;
        aoff[g] = (m < G::kM) ? co * G::kDS + (2 - ky) * G::kDyCols + (5 - kx) : COUT * G::kDS;   // else: the zero rows
    }

    f32x4 acc[MG];
#pragma unroll
    for (int g = 0; g < MG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* pad_in = g_pad_consts + (IN == IN_BNRELU ? 0 : 4);
    const float* pad_zero = g_pad_consts + 4;

    auto issue_dma = [&](int tile, int buf) {
        const int n = tile / tiles_per_sample;
        const int trem = tile - n * tiles_per_sample;
        const int x0 = (trem % p.tiles_x) * G::kTX;
        const int y0 = (trem / p.tiles_x) * G::kTY;
        const WgSample sm(p, n);
        const float* in_s = p.in + sm.in_off(p);
        const float* dy_s = p.dy + sm.dy_off(p);
        float* s_in = smem + buf * G::kBuf;
        float* s_dy = s_in + 16 * G::kCS;
        if constexpr (IN == IN_UPSAMPLE) {
            // nearest x2 gather: one dword per output pixel, 4 issues per channel; wave w moves channels w, w+4, ...
#pragma unroll
            for (int k = 0; k < 16 / G::kWaves; ++k) {
                const int c = wave + G::kWaves * k;
                const int ch = ci_base + c;
                const float* plane = in_s + static_cast<int64_t>(ch) * p.in_cs;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = q * 64 + lane;
                    const int ry = e / G::kTX, rx = e % G::kTX;
                    const int gy = y0 + ry, gx = x0 + rx;
                    const bool ok = gy < p.h && gx < p.w && ch < p.cin;
                    const float* src = ok ? plane + (gy >> 1) * p.in_w + (gx >> 1) : pad_zero;
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_in + c * G::kCS + q * 64), 4, 0, 0);
                }
            }
        } else {   // activations: 16 channels x 64 float4, thread -> (row, 4-pixel group) of the tile
            const int e = tid & 63;            // unit inside a channel: wave w handles channels w, w+4, ...
            const int ry = e / (G::kTX / 4), rx = (e % (G::kTX / 4)) * 4;
            const int gy = y0 + ry, gx = x0 + rx;
            const bool ok = gy < p.h && gx < p.w;
            const float* base = in_s + gy * p.in_w + gx;
#pragma unroll
            for (int k = 0; k < 16 / G::kWaves; ++k) {
                const int c = wave + G::kWaves * k;
                const int ch = ci_base + c;
                const float* src = (ok && ch < p.cin) ? base + static_cast<int64_t>(ch) * p.in_cs : pad_in;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_in + c * G::kCS), 16, 0, 0);
            }
        }
        // dY: COUT maps x 100 float4 (1-row / 4-column halo); wave w moves maps w, w+4, ... (two issues per map)
#pragma unroll
        for (int k = 0; k < (COUT + G::kWaves - 1) / G::kWaves; ++k) {
            const int co = wave + G::kWaves * k;
            if (co < COUT && co_base + co < p.cout) {
                const float* map = dy_s + static_cast<int64_t>(co_base + co) * p.dy_cs;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int u = half * 64 + lane;
                    if (u < G::kDyUnits) {
                        const int ry = u / (G::kDyCols / 4), rx = (u % (G::kDyCols / 4)) * 4;
                        const int gy = y0 - 1 + ry, gx = x0 - 4 + rx;
                        const bool ok = gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
                        const float* src = ok ? map + gy * p.dy_w + gx : pad_zero;
                        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_dy + co * G::kDS + half * 256), 16, 0, 0);
                    }
                }
            }
        }
    };

    // the zero rows of both buffers (never touched by the DMA)
    for (int i = tid; i < G::kZero; i += 64 * G::kWaves) {
        smem[16 * G::kCS + COUT * G::kDS + i] = 0.f;
        smem[G::kBuf + 16 * G::kCS + COUT * G::kDS + i] = 0.f;
    }

    const int rows_per_wave = G::kTY / G::kWaves;      // 1

    int tile = group;
    if (tile < tiles_total) issue_dma(tile, 0);
    int it = 0;
    for (; tile < tiles_total; tile += gridDim.y, ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int next = tile + gridDim.y;
        if (next < tiles_total) issue_dma(next, (it + 1) & 1);

        const float* s_in = smem + (it & 1) * G::kBuf;
        const float* s_dy = s_in + 16 * G::kCS;
        const float* cst = s_cst + 64 * WgSample(p, tile / tiles_per_sample).grp;
        const float bsc = cst[li], bmn = cst[16 + li], bbt = cst[32 + li];
#pragma unroll
        for (int rr = 0; rr < rows_per_wave; ++rr) {
            const int row = wave * rows_per_wave + rr;
            if constexpr (BF != 0) {
                static_assert(G::kTX % 16 == 0, "whole groups of four k-steps");
#pragma unroll
                for (int xg = 0; xg < G::kTX / 16; ++xg) {
                    float b[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        b[i] = s_in[li * G::kCS + row * G::kTX + (4 * xg + i) * 4 + lk];
                        if constexpr (IN == IN_BNRELU) b[i] = __builtin_fmaxf(fmaf(b[i] - bmn, bsc, bbt), 0.f);
                    }
                    const bf16x4_bits bp = pack_bf16x4(b[0], b[1], b[2], b[3]);
                    const int abase = row * G::kDyCols + 16 * xg + lk;
#pragma unroll
                    for (int g = 0; g < MG; ++g) {
                        const float* ap = s_dy + aoff[g] + abase;
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pack_bf16x4(ap[0], ap[4], ap[8], ap[12]), bp, acc[g], 0, 0, 0);
                    }
                }
            } else
#pragma unroll 2
            for (int x4 = 0; x4 < G::kTX / 4; ++x4) {
                const int pix = row * G::kTX + x4 * 4 + lk;
                float b = s_in[li * G::kCS + pix];
                if constexpr (IN == IN_BNRELU) b = __builtin_fmaxf(fmaf(b - bmn, bsc, bbt), 0.f);
                const int abase = row * G::kDyCols + x4 * 4 + lk;
#pragma unroll
                for (int g = 0; g < MG; ++g) {
                    const float a = s_dy[aoff[g] + abase];
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g], 0, 0, 0);
                }
            }
        }
    }

    // ---- cross-wave reduction through LDS, then one atomic per element ----
    // lane holds D[m = 16 g + 4 lk + e][ci = li]
    __syncthreads();
    float* s_red = smem;
#pragma unroll
    for (int g = 0; g < MG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) s_red[((wave * MG + g) * 4 + e) * 64 + lane] = acc[g][e];
    __syncthreads();
    for (int idx = tid; idx < MG * 256; idx += 64 * G::kWaves) {
        const int ln = idx & 63;
        const int e = (idx >> 6) & 3;
        const int g = idx >> 8;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < G::kWaves; ++wv) v += s_red[wv * MG * 256 + idx];
        const int m = 16 * g + 4 * (ln >> 4) + e;
        const int ci = ci_base + (ln & 15);
        if (m < G::kM && ci < p.cin) {
            const int co = co_base + m / 9, tap = m % 9;
            if (co < p.cout) atomicAdd(p.dw + (static_cast<int64_t>(co) * p.cin + ci) * 9 + tap, v);
        }
    }
}

template <int COUT, int IN, int BF = 0>
inline int launch_wgrad_taps(WgradParams p, hipStream_t stream) {
    using G = WgradTapsGeom<COUT>;
    p.tiles_x = (p.w + G::kTX - 1) / G::kTX;
    p.tiles_y = (p.h + G::kTY - 1) / G::kTY;
    const int ci_chunks = (p.cin + 15) / 16;
    const int co_sets = (p.cout + COUT - 1) / COUT;
    const int tiles_total = p.tiles_x * p.tiles_y * p.n;
    // 2 blocks fit a CU (LDS): aim at exactly one round of 512 equally loaded blocks
    int groups = 512 / (ci_chunks * co_sets);
    if (groups < 1) groups = 1;
    if (groups > tiles_total) groups = tiles_total;
    if (groups >= 16) groups &= ~7;            // multiple of 8: enables the XCD-local block order
    static bool configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& configured = configured_by_device[dev & 15];
    if (!configured && G::kBytes > 48 * 1024) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_taps_kernel<COUT, IN, BF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(G::kBytes)));
        configured = true;
    }
    wgrad_taps_kernel<COUT, IN, BF><<<dim3(ci_chunks, groups, co_sets), 64 * G::kWaves, G::kBytes, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// upsampled = the activation operand is gathered (dword DMA), only dY needs float4 alignment
inline bool wgrad_taps_ok(const WgradParams& p, bool upsampled = false) {
    const bool dy_ok = (p.w % 4 == 0) && (p.dy_w % 4 == 0) && (p.dy_cs % 4 == 0) && (p.dy_ns % 4 == 0) &&
                       (reinterpret_cast<uintptr_t>(p.dy) % 16 == 0);
    const bool in_ok = upsampled || ((p.in_w % 4 == 0) && (p.in_cs % 4 == 0) && (p.in_ns % 4 == 0) && (reinterpret_cast<uintptr_t>(p.in) % 16 == 0));
    return dy_ok && in_ok;
}

}  // namespace endo
