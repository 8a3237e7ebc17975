// Weight gradients on the fp32 matrix cores: dW[co][ci][tap] += sum_pixels dY[co][p] * a[ci][p + tap].
//
// GEMM view: M = cout (16 per MFMA, 12 used for the growth-12 layers), N = 16 input channels,
// K = pixels (4 consecutive x per MFMA).  A block owns one 16-channel input slice (and one set of
// 16*Q output channels) and walks a strided subset of the 32 x 8 pixel tiles, keeping the 9*Q
// accumulators in registers across tiles; at the end the 4 waves are summed through LDS and the
// result is added to dW with one fp32 atomic per element per block.  The input tile goes through
// the same fused load path as the forward convolution (BN+ReLU from the saved batch statistics /
// nearest-x2 gather), so activations are never materialised.
// MFMA roles: A[i = cout][k = pixel] from the dY tile, B[k = pixel][j = cin] from the input tile;
// both LDS images use a channel stride == 2 (mod 32) dwords, which makes the (16 channels x 2
// pixels) footprint of each 32-lane group bank-conflict free.
#pragma once

#include "conv_kernels.h"

namespace endo {

constexpr int kWgTileX = 32;
constexpr int kWgTileY = 8;
constexpr int kWgKC = 16;

enum DyMode { DY_PLAIN = 0, DY_UNPOOL = 1 };

struct WgradParams {
    int n, h, w;
    int tiles_x, tiles_y;
    // activations feeding the conv
    const float* in;
    int64_t in_ns;
    int in_cs, in_w;
    int cin;
    const float* saved;      // BNRELU: [cin][2] mean, rstd
    const float* gamma;
    const float* beta;
    // output gradient
    const float* dy;
    int64_t dy_ns;
    int dy_cs, dy_w;
    const uint8_t* dy_idx;   // DY_UNPOOL
    int64_t idx_ns;
    int cout;
    float* dw;               // [cout][cin][KS*KS], accumulated
    // grouped batch (see ConvParams): n counts groups * group_n samples; sample s belongs to group s / group_n, whose
    // dy / dy_idx / saved live gs floats further on and whose activations in_gs floats further on.  dw sums over all groups.
    int group_n;
    int64_t gs, in_gs;
    // wgrad_f34_kernel<.., RAW> with prep_x set (the network's first convolution, the last kernel of the backward pass): `dy` holds the RAW
    // gradient d of the convolution's output channels and the kernel forms the prepared gradient G = d + P x + Q itself (x = the channels'
    // forward values, planes laid out as dy's; P, Q = the channels' deferred BatchNorm-backward terms, per group gs floats apart) and adds
    // sum G into the bias gradient -- what prep_dy_kernel does in a pass of its own (3 x 48 planes of traffic with nothing else on the chip)
    const float* prep_x;
    const float* prep_p;
    const float* prep_q;
    float* prep_bias;
};

constexpr int kMaxGroups = 4;

// group and in-group index of global sample s; offsets of its planes
struct WgSample {
    int grp, nl;
    __device__ __forceinline__ WgSample(const WgradParams& p, int s) {
        grp = p.group_n > 0 ? s / p.group_n : 0;
        nl = s - grp * p.group_n;
    }
    __device__ __forceinline__ int64_t in_off(const WgradParams& p) const { return grp * p.in_gs + nl * p.in_ns; }
    __device__ __forceinline__ int64_t dy_off(const WgradParams& p) const { return grp * p.gs + nl * p.dy_ns; }
    __device__ __forceinline__ int64_t idx_off(const WgradParams& p) const { return 4 * grp * p.gs + nl * p.idx_ns; }
};
__device__ __forceinline__ int wg_groups(const WgradParams& p) { return p.group_n > 0 ? p.n / p.group_n : 1; }

template <int KS>
struct WgradGeom {
    static constexpr int kHalo = KS / 2;
    static constexpr int kRows = kWgTileY + 2 * kHalo;
    static constexpr int kCols = kWgTileX + 2 * kHalo;
    static constexpr int kPlane = kRows * kCols;
    static constexpr int kCS = ((kPlane - 2 + 31) / 32) * 32 + 2;       // == 2 (mod 32)
    static constexpr int kDyPlane = kWgTileX * kWgTileY;
    static constexpr int kDS = kDyPlane + 2;                            // 258 == 2 (mod 32)
    static constexpr int kPos = (kPlane + kConvThreads - 1) / kConvThreads;
    static constexpr int kPre = kWgKC * kPos;
};

template <int KS, int Q, int IN, int DY>
__global__ void __launch_bounds__(kConvThreads) wgrad_mfma_kernel(const WgradParams p) {
    using G = WgradGeom<KS>;
    constexpr int KK = KS * KS;
    constexpr int NB = 16 * Q;
    constexpr int kDyElems = NB * G::kDyPlane;
    constexpr int kDyPre = kDyElems / kConvThreads;
    constexpr int RW = kWgTileY / 4;     // rows per wave

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;                        // [16][kCS]
    float* s_dy = s_in + kWgKC * G::kCS;       // [NB][kDS]
    __shared__ float s_cst[kMaxGroups * 4 * kWgKC];   // per group: scale, mean, beta per input channel

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int ci_base = blockIdx.x * kWgKC;
    const int co_base = blockIdx.z * NB;
    const int tiles_per_sample = p.tiles_x * p.tiles_y;
    const int tiles_total = tiles_per_sample * p.n;

    if constexpr (IN == IN_BNRELU) {
        if (tid < kWgKC * wg_groups(p)) {
            const int g = tid / kWgKC, t = tid - g * kWgKC;
            const int c = ci_base + t;
            float scale = 0.f, mean = 0.f, beta = 0.f;
            if (c < p.cin) {
                const float* saved = p.saved + g * p.gs;
                mean = saved[2 * c];
                scale = p.gamma[c] * saved[2 * c + 1];
                beta = p.beta[c];
            }
            s_cst[g * 4 * kWgKC + t] = scale;
            s_cst[g * 4 * kWgKC + kWgKC + t] = mean;
            s_cst[g * 4 * kWgKC + 2 * kWgKC + t] = beta;
        }
    }

    f32x4 acc[KK][Q];
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    float pre[G::kPre];
    float dpre[kDyPre];
    unsigned pos_ok = 0;     // validity of this thread's tile positions for the tile held in `pre`

    // thread-fixed decomposition of tile positions (input tile: kPos per thread, dY tile: 1 per thread)
    int in_ry[G::kPos], in_rx[G::kPos];
#pragma unroll
    for (int k = 0; k < G::kPos; ++k) {
        const int e = tid + k * kConvThreads;
        in_ry[k] = e / G::kCols;
        in_rx[k] = e - in_ry[k] * G::kCols;
    }
    const int dy_ry = tid / kWgTileX, dy_rx = tid % kWgTileX;
    static_assert(kWgTileX * kWgTileY == kConvThreads, "one dY pixel per thread");
    static_assert(kDyPre == NB, "dY staging is one value per thread per output channel");

    int pre_grp = 0;         // group of the tile held in `pre`
    auto load_tile = [&](int tile) {
        const int n = tile / tiles_per_sample;
        const int trem = tile - n * tiles_per_sample;
        const int x0 = (trem % p.tiles_x) * kWgTileX;
        const int y0 = (trem / p.tiles_x) * kWgTileY;
        const WgSample sm(p, n);
        pre_grp = sm.grp;
        const float* in_n = p.in + sm.in_off(p);
        pos_ok = 0;
        int goff[G::kPos];
#pragma unroll
        for (int k = 0; k < G::kPos; ++k) {
            const int gy = y0 - G::kHalo + in_ry[k];
            const int gx = x0 - G::kHalo + in_rx[k];
            goff[k] = 0;
            if (tid + k * kConvThreads < G::kPlane && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) {
                pos_ok |= (1u << k);
                if constexpr (IN == IN_UPSAMPLE) goff[k] = (gy >> 1) * p.in_w + (gx >> 1);
                else goff[k] = gy * p.in_w + gx;
            }
        }
#pragma unroll
        for (int c = 0; c < kWgKC; ++c) {
            const int ch = ci_base + c;
            const int64_t coff = static_cast<int64_t>(ch) * p.in_cs;
#pragma unroll
            for (int k = 0; k < G::kPos; ++k)
                pre[c * G::kPos + k] = (ch < p.cin && (pos_ok & (1u << k))) ? in_n[coff + goff[k]] : 0.f;
        }
        const int gy = y0 + dy_ry, gx = x0 + dy_rx;
        const bool ok = gy < p.h && gx < p.w;
        const float* dy_n = p.dy + sm.dy_off(p);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int co = co_base + j;
            float v = 0.f;
            if (ok && co < p.cout) {
                if constexpr (DY == DY_UNPOOL) {
                    const int64_t o = static_cast<int64_t>(co) * p.dy_cs + (gy >> 1) * p.dy_w + (gx >> 1);
                    const int code = ((gy & 1) << 1) | (gx & 1);
                    v = (p.dy_idx[sm.idx_off(p) + o] == code) ? dy_n[o] : 0.f;
                } else {
                    v = dy_n[static_cast<int64_t>(co) * p.dy_cs + gy * p.dy_w + gx];
                }
            }
            dpre[j] = v;
        }
    };

    auto store_tile = [&]() {
        const float* cst = s_cst + pre_grp * 4 * kWgKC;
#pragma unroll
        for (int c = 0; c < kWgKC; ++c) {
#pragma unroll
            for (int k = 0; k < G::kPos; ++k) {
                const int e = tid + k * kConvThreads;
                if (e < G::kPlane) {
                    float v = pre[c * G::kPos + k];
                    if constexpr (IN == IN_BNRELU) {
                        if (ci_base + c < p.cin && (pos_ok & (1u << k))) {
                            v = fmaf(v - cst[kWgKC + c], cst[c], cst[2 * kWgKC + c]);
                            v = v > 0.f ? v : 0.f;
                        } else {
                            v = 0.f;
                        }
                    }
                    s_in[c * G::kCS + e] = v;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) s_dy[j * G::kDS + tid] = dpre[j];
    };

    int tile = blockIdx.y;
    if (tile < tiles_total) load_tile(tile);
    __syncthreads();   // s_cst
    bool first = true;
    for (; tile < tiles_total; tile += gridDim.y) {
        if (!first) __syncthreads();
        first = false;
        store_tile();
        __syncthreads();
        if (tile + static_cast<int>(gridDim.y) < tiles_total) load_tile(tile + gridDim.y);

#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int row = wave * RW + rr;
#pragma unroll 2
            for (int x4 = 0; x4 < kWgTileX / 4; ++x4) {
                float a[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) a[q] = s_dy[(q * 16 + li) * G::kDS + row * kWgTileX + x4 * 4 + lk];
#pragma unroll
                for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx) {
                        const float b = s_in[li * G::kCS + (row + dy) * G::kCols + x4 * 4 + lk + dx];
#pragma unroll
                        for (int q = 0; q < Q; ++q)
                            acc[dy * KS + dx][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b, acc[dy * KS + dx][q], 0, 0, 0);
                    }
            }
        }
    }

    // ---- cross-wave reduction through LDS, then one atomic per element ----
    // lane holds D[i = co 4*lk+e][j = ci li] for every (tap, q)
    float* s_red = smem;    // [4 waves][KK][4][64]
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < KK; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) s_red[((wave * KK + t) * 4 + e) * 64 + lane] = acc[t][q][e];
        __syncthreads();
        for (int idx = tid; idx < KK * 4 * 64; idx += kConvThreads) {
            const int ln = idx & 63;
            const int e = (idx >> 6) & 3;
            const int t = idx >> 8;
            const float v = s_red[idx] + s_red[KK * 256 + idx] + s_red[2 * KK * 256 + idx] + s_red[3 * KK * 256 + idx];
            const int co = co_base + q * 16 + 4 * (ln >> 4) + e;
            const int ci = ci_base + (ln & 15);
            if (co < p.cout && ci < p.cin) atomicAdd(p.dw + (static_cast<int64_t>(co) * p.cin + ci) * KK + t, v);
        }
    }
}

template <int KS, int Q>
constexpr size_t wgrad_smem_bytes() {
    using G = WgradGeom<KS>;
    size_t tiles = sizeof(float) * (kWgKC * G::kCS + 16 * Q * G::kDS);
    size_t red = sizeof(float) * 4 * KS * KS * 256;
    return tiles > red ? tiles : red;
}

template <int KS, int Q, int IN, int DY>
inline int launch_wgrad(const WgradParams& p, hipStream_t stream) {
    const int ci_chunks = (p.cin + kWgKC - 1) / kWgKC;
    const int co_sets = (p.cout + 16 * Q - 1) / (16 * Q);
    const int tiles_total = p.tiles_x * p.tiles_y * p.n;
    // ~1024 blocks in total; at most 384 blocks add into the same dW element (atomic contention)
    int groups = 1024 / (ci_chunks * co_sets);
    if (groups < 1) groups = 1;
    if (groups > 384) groups = 384;
    if (groups > tiles_total) groups = tiles_total;
    dim3 grid(ci_chunks, groups, co_sets);
    constexpr size_t smem = wgrad_smem_bytes<KS, Q>();
    static bool configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& configured = configured_by_device[dev & 15];
    if (!configured && smem > 48 * 1024) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_mfma_kernel<KS, Q, IN, DY>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
        configured = true;
    }
    wgrad_mfma_kernel<KS, Q, IN, DY><<<grid, kConvThreads, smem, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// 1x1 weight gradient (transition down): dW[co][ci] += sum_p dY[co][p] * a[ci][p], a plain GEMM with
// K = all pixels.  The 3x3 kernel above re-stages a tile for 16 MFMAs per wave; here a block owns a
// 96 x 96 (co x ci) tile of dW, streams 64-pixel chunks of both operands through LDS (dY un-pooled on
// the fly from the pooled gradient + argmax codes, a = relu(bn(x)) from the saved statistics) and
// issues 144 MFMAs per wave per chunk: wave (wr, wc) owns 3 x 3 sixteen-wide sub-tiles.
// ---------------------------------------------------------------------------------------------
constexpr int kW1Tile = 96;
constexpr int kW1Chunk = 64;
constexpr int kW1Stride = kW1Chunk + 2;     // == 2 (mod 32): conflict-free (16 channels x 2 pixels) reads
constexpr int kW1Rows = 2 * kW1Tile;        // 96 dY rows then 96 activation rows
constexpr int kW1Pre = kW1Rows * kW1Chunk / kConvThreads;   // 48 staged values per thread

__global__ void __launch_bounds__(kConvThreads) wgrad1x1_mfma_kernel(const WgradParams p) {
    __shared__ float s_t[kW1Rows * kW1Stride];
    __shared__ float s_cst[kMaxGroups * 3 * kW1Tile];       // per group: scale, mean, beta
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int co_base = blockIdx.y * kW1Tile;
    const int ci_base = blockIdx.z * kW1Tile;
    const int plane = p.h * p.w;
    const int chunks_per_sample = (plane + kW1Chunk - 1) / kW1Chunk;
    const int chunks_total = chunks_per_sample * p.n;

    for (int e = tid; e < kW1Tile * wg_groups(p); e += kConvThreads) {
        const int g = e / kW1Tile, c = e - g * kW1Tile;
        const int ch = ci_base + c;
        float scale = 0.f, mean = 0.f, beta = 0.f;
        if (ch < p.cin) {
            const float* saved = p.saved + g * p.gs;
            mean = saved[2 * ch];
            scale = p.gamma[ch] * saved[2 * ch + 1];
            beta = p.beta[ch];
        }
        float* cst = s_cst + g * 3 * kW1Tile;
        cst[c] = scale; cst[kW1Tile + c] = mean; cst[2 * kW1Tile + c] = beta;
    }

    f32x4 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int px = tid & 63;            // pixel inside the chunk this thread stages
    const int row0 = tid >> 6;          // rows row0 + 4k
    float pre[kW1Pre];
    bool pix_ok = false;
    int pre_grp = 0;

    auto load_chunk = [&](int chunk) {
        const int n = chunk / chunks_per_sample;
        const WgSample sm(p, n);
        pre_grp = sm.grp;
        const int pix = (chunk - n * chunks_per_sample) * kW1Chunk + px;
        pix_ok = pix < plane;
        int pooled = 0, code = 0;
        if (pix_ok) {
            const int y = pix / p.w, x = pix - y * p.w;
            pooled = (y >> 1) * p.dy_w + (x >> 1);
            code = ((y & 1) << 1) | (x & 1);
        }
        const float* dy_n = p.dy + sm.dy_off(p);
        const uint8_t* idx_n = p.dy_idx + sm.idx_off(p);
        const float* in_n = p.in + sm.in_off(p);
#pragma unroll
        for (int k = 0; k < kW1Pre / 2; ++k) {          // dY rows
            const int co = co_base + row0 + 4 * k;
            float v = 0.f;
            if (pix_ok && co < p.cout) {
                const int64_t o = static_cast<int64_t>(co) * p.dy_cs + pooled;
                v = (idx_n[o] == code) ? dy_n[o] : 0.f;
            }
            pre[k] = v;
        }
#pragma unroll
        for (int k = 0; k < kW1Pre / 2; ++k) {          // activation rows
            const int ch = ci_base + row0 + 4 * k;
            pre[kW1Pre / 2 + k] = (pix_ok && ch < p.cin) ? in_n[static_cast<int64_t>(ch) * p.in_cs + pix] : 0.f;
        }
    };
    auto store_chunk = [&]() {
        const float* cst = s_cst + pre_grp * 3 * kW1Tile;
#pragma unroll
        for (int k = 0; k < kW1Pre / 2; ++k) s_t[(row0 + 4 * k) * kW1Stride + px] = pre[k];
#pragma unroll
        for (int k = 0; k < kW1Pre / 2; ++k) {
            const int r = row0 + 4 * k;
            float v = 0.f;
            if (pix_ok && ci_base + r < p.cin) {
                v = fmaf(pre[kW1Pre / 2 + k] - cst[kW1Tile + r], cst[r], cst[2 * kW1Tile + r]);
                v = v > 0.f ? v : 0.f;
            }
            s_t[(kW1Tile + r) * kW1Stride + px] = v;
        }
    };

    const int wr = wave >> 1, wc = wave & 1;      // wave's 48 x 48 quadrant of the 96 x 96 tile
    int chunk = blockIdx.x;
    if (chunk < chunks_total) load_chunk(chunk);
    __syncthreads();
    bool first = true;
    for (; chunk < chunks_total; chunk += gridDim.x) {
        if (!first) __syncthreads();
        first = false;
        store_chunk();
        __syncthreads();
        if (chunk + static_cast<int>(gridDim.x) < chunks_total) load_chunk(chunk + gridDim.x);
        const float* a_base = s_t + (wr * 48 + li) * kW1Stride + lk;
        const float* b_base = s_t + (kW1Tile + wc * 48 + li) * kW1Stride + lk;
#pragma unroll 4
        for (int ks = 0; ks < kW1Chunk / 4; ++ks) {
            float a[3], b[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                a[i] = a_base[i * 16 * kW1Stride + ks * 4];
                b[i] = b_base[i * 16 * kW1Stride + ks * 4];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // lane holds D[co = 4*lk + e][ci = li] of each 16 x 16 sub-tile
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co_base + wr * 48 + i * 16 + 4 * lk + e;
                const int ci = ci_base + wc * 48 + j * 16 + li;
                if (co < p.cout && ci < p.cin) atomicAdd(p.dw + static_cast<int64_t>(co) * p.cin + ci, acc[i][j][e]);
            }
}

inline int launch_wgrad1x1(const WgradParams& p, hipStream_t stream) {
    const int tiles_co = (p.cout + kW1Tile - 1) / kW1Tile;
    const int tiles_ci = (p.cin + kW1Tile - 1) / kW1Tile;
    const int plane = p.h * p.w;
    const int chunks_total = ((plane + kW1Chunk - 1) / kW1Chunk) * p.n;
    int splits = 768 / (tiles_co * tiles_ci);
    if (splits < 1) splits = 1;
    if (splits > chunks_total) splits = chunks_total;
    wgrad1x1_mfma_kernel<<<dim3(splits, tiles_co, tiles_ci), kConvThreads, 0, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

}  // namespace endo
