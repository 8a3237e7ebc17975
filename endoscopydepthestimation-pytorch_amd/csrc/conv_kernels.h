// Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), NCHW planar.
//
// One kernel template covers every 3x3 / 1x1 convolution of FC-DenseNet57 forward AND the data
// gradients of backward (a dgrad is the same convolution with the weight tensor transposed and
// flipped).  What differs per use is fused into the load path and the epilogue:
//
//   load path   PLAIN      raw values                                   (first conv, dgrad inputs)
//               BNRELU     relu(scale[c] * x + shift[c]); scale/shift are derived in the block
//                          prologue from the per-channel batch sums (training) or running stats
//               UPSAMPLE   nearest x2 gather from a half-resolution plane (transition up)
//               UNPOOL     routes a pooled gradient back through the stored 2x2 argmax (1x1 only)
//   epilogue    FWD        + bias, store into a channel slice of the level buffer (this replaces
//                          torch.cat), accumulate per-channel sum / sum^2 for later BN layers
//               FWD_POOL   + bias, 2x2 max-pool in registers, store pooled value + argmax, stats
//               DGRAD_BN   ReLU mask + BN backward (training mode) fused: dz = da * [z > 0];
//                          sum dz and sum dz*xhat -> fp64 atomics; dbuf (+)= gamma*rstd * dz.  The
//                          mean-subtraction terms of BN backward are affine in x per channel and
//                          are applied lazily (see small_kernels: bn_bwd_finalize / prep_dy).
//               DGRAD_SUMPOOL  2x2 sum (backward of nearest x2) and plain store
//
// Tiling: a block of 4 waves computes a 32 x 16 pixel tile for NB = 16*Q output channels.  Wave w
// owns a 16-pixel-wide, 8-row strip; per (input-channel quad, kx) it reads 8+2 A fragments and
// 3*Q B fragments from LDS and issues 24*Q MFMAs (the three ky taps reuse the A fragments by row
// shift).  MFMA roles: A[i = pixel x][k = channel], B[k = channel][j = cout]; D[i][j] lands with
// 4 consecutive x pixels per lane for cout j = lane & 15, so stores are float4 along W.
// LDS: input tile [KC][rows][cols] with channel stride == 16 (mod 32) dwords and weight tile
// [tap][KC][NB]; both fragment reads are bank-conflict free for ds_read_b32 (32-lane groups).
// Global->LDS staging is register-prefetched one K-chunk ahead so HBM latency hides under MFMAs.
#pragma once

#include "common.h"

namespace endo {

enum InMode { IN_PLAIN = 0, IN_BNRELU = 1, IN_UPSAMPLE = 2, IN_UNPOOL = 3, IN_SUBPIX = 4 };
enum Epilogue { EPI_FWD = 0, EPI_FWD_POOL = 1, EPI_DGRAD_BN = 2, EPI_DGRAD_SUMPOOL = 3 };

constexpr int kConvThreads = 256;
constexpr int kMaxBnChannels = 384;

struct ConvParams {
    int n, h, w;      // output grid of the convolution (before any pooling in the epilogue)
    int tiles_x;
    // ---- input ----
    const float* in;
    int64_t in_ns;    // sample stride (floats)
    int in_cs;        // channel-plane stride
    int in_w;         // row stride
    const uint8_t* in_idx;   // UNPOOL: argmax codes, same indexing as `in`
    int64_t idx_ns;
    int cin;          // logical input channels of this launch
    // BNRELU on the input channels
    const double* in_sums;   // [cin][2] sum, sum^2 (training)
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* saved;            // [cin][2] mean, rstd written by block 0 for backward
    double count;            // N*H*W of the normalised tensor
    float eps, momentum;
    int training;
    // ---- weights: the ORIGINAL conv's tensor [w_cout][w_cin][KS][KS] ----
    const float* wgt;
    const float* bias;
    int w_cout, w_cin;
    // the same weights already in the K-chunk pipeline's LDS order, [chunk of 16 input channels][tap 9][channel 16][cout 16], zeros where a
    // channel or an output does not exist (dense_fwd_weights: wino_fwd_kernels.h, mode 1): one contiguous 9 KB run per chunk instead of
    // 2 304 four-byte gathers.  Only the KS = 3, KC = 16, Q = 1 forward instantiation of conv_dma_kernel looks at it; nullptr = gather from wgt
    const float* wgt_chunks;
    // ---- output ----
    float* out;
    int64_t out_ns;
    int out_cs, out_w;
    int cout;         // logical output channels of this launch
    uint8_t* out_idx;        // FWD_POOL argmax codes (indexing as `out`)
    double* out_sums;        // [cout][2] FWD / FWD_POOL statistics of the stored values
    // ---- DGRAD_BN ----
    const float* x;          // activations of the output channels (same resolution)
    int64_t x_ns;
    int x_cs;
    const float* bn_saved;   // [cout][2] mean, rstd
    const float* bn_gamma;
    const float* bn_beta;
    double* bn_scratch;      // [cout][2] sum dz, sum dz*xhat
    int64_t bn_slot_stride;  // copies of bn_scratch, this many doubles apart (common.h: kBnSlots); 0 = one copy
    int acc_from;            // output channels >= acc_from accumulate into `out`, others overwrite
    int bn_cap;              // LDS-DMA kernels: capacity (channels) of the BN constant tables, set by the launcher
    // split-K (coarse levels): blockIdx.y = slice of the input channels; slice s writes its raw partial sums at
    // out + s * split_stride (bias / statistics are applied by finalize_partial_kernel)
    int ksplit;
    int64_t split_stride;
    // ---- grouped batch (endo_net_create_grouped): blockIdx.z runs over groups * group_n samples.  Group g is an
    // independent forward / backward (its own BatchNorm statistics, tape and gradient workspace) that shares the
    // parameters: tape / workspace pointers move by g * gs floats, `in` and `out` by their own strides (they may be
    // caller tensors), parameter pointers do not move.  group_n == 0 means "not grouped" (one group of n samples).
    int group_n;
    int64_t gs, in_gs, out_gs;
    int sub_c;        // IN_SUBPIX: real channels per sub-pixel phase (cin = 4 * sub_c pseudo-channels)
    // ---- the last dense layer of the network (wino4_fwd_kernel<true>): the final 1x1 convolution over the channels this launch streams
    // anyway -- fin_out[sample][pixel] = sum over the launch's input channels of fin_w[c] * x_raw[c][pixel] (no BN, no ReLU: finalConv reads the
    // raw concatenation, reference models.py:167, 186).  fin_out is a tape pointer: it moves with the group like `out`'s tape does (gs).
    const float* fin_w;
    float* fin_out;
};

// this block's group and sample inside it
__device__ __forceinline__ void group_of(const ConvParams& p, int z, int& grp, int& n) {
    grp = p.group_n > 0 ? z / p.group_n : 0;
    n = z - grp * p.group_n;
}

__device__ __forceinline__ ConvParams group_view(const ConvParams& p, int grp) {
    ConvParams q = p;
    const int64_t o = grp * p.gs;
    q.in += grp * p.in_gs;
    q.out += grp * p.out_gs;
    if (q.x) q.x += o;
    if (q.in_idx) q.in_idx += 4 * o;
    if (q.out_idx) q.out_idx += 4 * o;
    if (q.in_sums) q.in_sums += o / 2;
    if (q.out_sums) q.out_sums += o / 2;
    if (q.bn_scratch) q.bn_scratch += o / 2;
    if (q.saved) q.saved += o;
    if (q.bn_saved) q.bn_saved += o;
    return q;
}

// BN constants of input channel c for this block's group (p = the group view, p0 = the launch parameters).  The first
// block of each group records (mean, rstd) for its backward pass; the very first block of the launch applies the
// running-statistics updates of ALL groups, in group order -- the sequence the reference's separate forward calls
// (one per group) would produce.
__device__ __forceinline__ void bn_input_constants(const ConvParams& p, const ConvParams& p0, int grp, int groups, bool first_of_group,
                                                   int c, float& scale, float& mean_f, float& beta) {
    double mean, var;
    if (p.training) {
        mean = p.in_sums[2 * c] / p.count;
        var = p.in_sums[2 * c + 1] / p.count - mean * mean;
        if (var < 0.0) var = 0.0;
    } else {
        mean = p.running_mean[c];
        var = p.running_var[c];
    }
    const double rstd = 1.0 / sqrt(var + static_cast<double>(p.eps));
    scale = p.gamma[c] * static_cast<float>(rstd);
    mean_f = static_cast<float>(mean);
    beta = p.beta[c];
    if (first_of_group) {
        if (p.saved) {
            p.saved[2 * c] = static_cast<float>(mean);
            p.saved[2 * c + 1] = static_cast<float>(rstd);
        }
        if (p.training && grp == 0) {
            float rm = p.running_mean[c], rv = p.running_var[c];
            for (int g = 0; g < groups; ++g) {
                const double* sums = p0.in_sums + g * (p0.gs / 2);
                const double m = sums[2 * c] / p.count;
                double v = sums[2 * c + 1] / p.count - m * m;
                if (v < 0.0) v = 0.0;
                const double unbiased = p.count > 1.0 ? v * p.count / (p.count - 1.0) : v;
                rm = (1.0f - p.momentum) * rm + p.momentum * static_cast<float>(m);
                rv = (1.0f - p.momentum) * rv + p.momentum * static_cast<float>(unbiased);
            }
            p.running_mean[c] = rm;
            p.running_var[c] = rv;
        }
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Tile shape: WX waves side by side (16 pixels each), 4/WX waves stacked, R rows per wave.
//   <WX=2, R=8>  32 x 16 pixels  -- full-resolution levels
//   <WX=1, R=4>  16 x 16 pixels  -- mid levels (more blocks to fill 256 CUs)
//   <WX=1, R=2>  16 x  8 pixels  -- the 32x40 ... 8x10 levels
template <int KS, int KC, int WX, int R, int VEC = 1>
struct ConvGeom {
    static constexpr int kTileX = 16 * WX;
    static constexpr int kTileY = R * (4 / WX);
    static constexpr int kHalo = KS / 2;
    static constexpr int kRows = kTileY + 2 * kHalo;
    // VEC == 4 (16-byte LDS-DMA): the tile starts 4 pixels left of the output tile so that every
    // row is a whole number of aligned float4s; the 3 extra columns per side are never read
    static constexpr int kLeft = (VEC == 4 && KS == 3) ? 4 : kHalo;
    static constexpr int kCols = kTileX + 2 * kLeft;
    static constexpr int kColOff = kLeft - kHalo;          // fragment column = x - x0 + dx + kColOff
    static constexpr int kPlane = kRows * kCols;
    // channel stride == 16 (mod 32) so the 4 k-groups of an A fragment hit disjoint banks
    static constexpr int kCS = ((kPlane - 16 + 31) / 32) * 32 + 16;
    static constexpr int kUnits = kPlane / VEC;                                // DMA / staging units per channel
    static constexpr int kPos = (kUnits + kConvThreads - 1) / kConvThreads;   // units per thread
    static constexpr int kPre = KC * kPos;                                    // staged values per thread per chunk
};

// Epilogue shared by the register-staged and the LDS-DMA kernels.  `cst` holds the DGRAD_BN
// per-channel constants (4 per output channel of the block), `s_red` 8*NB floats of reduction scratch.
template <int Q, int EPI, int R>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x4 (&acc)[R][Q], const float* cst, float* s_red,
                                              int x0, int y0, int wx, int wy, int co_base, int n) {
    constexpr int NB = 16 * Q;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int lk = lane >> 4;
    (void)cst; (void)s_red; (void)wave;
    // lane holds, for cout j = co_base + q*16 + li, pixels x = x0 + wx + 4*lk + {0..3}, rows y0+wy+r
    const int px = x0 + wx + 4 * lk;
    const bool vec_ok = ((p.out_w & 3) == 0) && ((p.w & 3) == 0);

    if constexpr (EPI == EPI_FWD) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int co = co_base + q * 16 + li;
            const bool co_ok = co < p.cout;
            const float bias = (co_ok && p.bias) ? p.bias[co] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = y0 + wy + r;
                f32x4 v = acc[r][q];
                v[0] += bias; v[1] += bias; v[2] += bias; v[3] += bias;
                if (co_ok && y < p.h) {
                    float* dst = p.out + n * p.out_ns + static_cast<int64_t>(co) * p.out_cs + y * p.out_w + px;
                    if (vec_ok && px + 3 < p.w) {
                        *reinterpret_cast<f32x4*>(dst) = v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s1 += v[e]; s2 += v[e] * v[e]; }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (px + e < p.w) { dst[e] = v[e]; s1 += v[e]; s2 += v[e] * v[e]; }
                    }
                }
            }
            if (p.out_sums) {
                s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
                if (lk == 0) {
                    s_red[(wave * NB + q * 16 + li) * 2] = s1;
                    s_red[(wave * NB + q * 16 + li) * 2 + 1] = s2;
                }
            }
        }
        if (p.out_sums) {
            __syncthreads();
            if (tid < 2 * NB) {
                const int j = tid >> 1, which = tid & 1;
                if (co_base + j < p.cout) {
                    double t = 0.0;
                    for (int wv = 0; wv < 4; ++wv) t += static_cast<double>(s_red[(wv * NB + j) * 2 + which]);
                    atomicAdd(p.out_sums + 2 * (co_base + j) + which, t);
                }
            }
        }
    } else if constexpr (EPI == EPI_FWD_POOL) {
        const int ph = p.h >> 1, pw = p.w >> 1;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int co = co_base + q * 16 + li;
            const bool co_ok = co < p.cout;
            const float bias = (co_ok && p.bias) ? p.bias[co] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                const int yp = (y0 + wy + r) >> 1;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const int xp = (px + e) >> 1;
                    float best = acc[r][q][e] + bias;
                    int code = 0;
                    float v = acc[r][q][e + 1] + bias;
                    if (v > best) { best = v; code = 1; }
                    v = acc[r + 1][q][e] + bias;
                    if (v > best) { best = v; code = 2; }
                    v = acc[r + 1][q][e + 1] + bias;
                    if (v > best) { best = v; code = 3; }
                    if (co_ok && yp < ph && xp < pw) {
                        const int64_t o = static_cast<int64_t>(co) * p.out_cs + yp * p.out_w + xp;
                        p.out[n * p.out_ns + o] = best;
                        p.out_idx[n * p.idx_ns + o] = static_cast<uint8_t>(code);
                        s1 += best; s2 += best * best;
                    }
                }
            }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lk == 0) {
                s_red[(wave * NB + q * 16 + li) * 2] = s1;
                s_red[(wave * NB + q * 16 + li) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (p.out_sums && tid < 2 * NB) {          // no statistics in inference mode
            const int j = tid >> 1, which = tid & 1;
            if (co_base + j < p.cout) {
                double t = 0.0;
                for (int wv = 0; wv < 4; ++wv) t += static_cast<double>(s_red[(wv * NB + j) * 2 + which]);
                atomicAdd(p.out_sums + 2 * (co_base + j) + which, t);
            }
        }
    } else if constexpr (EPI == EPI_DGRAD_BN) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int jloc = q * 16 + li;
            const int co = co_base + jloc;
            const bool co_ok = co < p.cout;
            const float scale = cst[4 * jloc], beta = cst[4 * jloc + 1], mean = cst[4 * jloc + 2], rstd = cst[4 * jloc + 3];
            const bool accumulate = co >= p.acc_from;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = y0 + wy + r;
                if (co_ok && y < p.h) {
                    const int64_t xo = n * p.x_ns + static_cast<int64_t>(co) * p.x_cs + y * p.out_w + px;
                    float* dst = p.out + n * p.out_ns + static_cast<int64_t>(co) * p.out_cs + y * p.out_w + px;
                    if (vec_ok && px + 3 < p.w) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + xo);
                        f32x4 o = accumulate ? *reinterpret_cast<const f32x4*>(dst) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float xc = xv[e] - mean;
                            const float z = fmaf(xc, scale, beta);
                            const float dz = z > 0.f ? acc[r][q][e] : 0.f;
                            s1 += dz;
                            s2 += dz * (xc * rstd);
                            o[e] += scale * dz;
                        }
                        *reinterpret_cast<f32x4*>(dst) = o;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (px + e < p.w) {
                                const float xv = p.x[xo + e];
                                const float xc = xv - mean;
                                const float z = fmaf(xc, scale, beta);
                                const float dz = z > 0.f ? acc[r][q][e] : 0.f;
                                s1 += dz;
                                s2 += dz * (xc * rstd);
                                dst[e] = (accumulate ? dst[e] : 0.f) + scale * dz;
                            }
                        }
                    }
                }
            }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lk == 0) {
                s_red[(wave * NB + jloc) * 2] = s1;
                s_red[(wave * NB + jloc) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < 2 * NB) {
            const int j = tid >> 1, which = tid & 1;
            if (co_base + j < p.cout) {
                double t = 0.0;
                for (int wv = 0; wv < 4; ++wv) t += static_cast<double>(s_red[(wv * NB + j) * 2 + which]);
                atomicAdd(p.bn_scratch + bn_slot_offset(p.bn_slot_stride) + 2 * (co_base + j) + which, t);
            }
        }
    } else {   // EPI_DGRAD_SUMPOOL
        const int ph = p.h >> 1, pw = p.w >> 1;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int co = co_base + q * 16 + li;
            if (co >= p.cout) continue;
#pragma unroll
            for (int r = 0; r < R; r += 2) {
                const int yp = (y0 + wy + r) >> 1;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const int xp = (px + e) >> 1;
                    if (yp < ph && xp < pw) {
                        const float v = (acc[r][q][e] + acc[r][q][e + 1]) + (acc[r + 1][q][e] + acc[r + 1][q][e + 1]);
                        p.out[n * p.out_ns + static_cast<int64_t>(co) * p.out_cs + yp * p.out_w + xp] = v;
                    }
                }
            }
        }
    }
}

template <int KS, int KC, int Q, int IN, int EPI, int WX, int R>
__global__ void __launch_bounds__(kConvThreads) conv_mfma_kernel(const ConvParams p0) {
    using G = ConvGeom<KS, KC, WX, R>;
    constexpr int kTileX = G::kTileX;
    constexpr int kTileY = G::kTileY;
    constexpr int KK = KS * KS;
    constexpr int NB = 16 * Q;
    constexpr int kWElems = KK * KC * NB;
    constexpr int kWPre = (kWElems + kConvThreads - 1) / kConvThreads;
    constexpr bool kDgrad = (EPI == EPI_DGRAD_BN || EPI == EPI_DGRAD_SUMPOOL);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;                       // [KC][kCS]
    float* s_w = s_in + KC * G::kCS;          // [KK][KC][NB]
    float* s_aux = s_w + kWElems;             // BN scale/shift or dgrad constants / reductions

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tile = blockIdx.x;
    const int x0 = (tile % p0.tiles_x) * kTileX;
    const int y0 = (tile / p0.tiles_x) * kTileY;
    const int co_base = blockIdx.y * NB;
    int grp, n;
    group_of(p0, blockIdx.z, grp, n);
    const ConvParams p = group_view(p0, grp);
    const int groups = p0.group_n > 0 ? gridDim.z / p0.group_n : 1;
    const bool first_of_group = blockIdx.x == 0 && blockIdx.y == 0 && n == 0;
    (void)groups; (void)first_of_group;

    // ---------------- prologue: per-channel constants ----------------
    if constexpr (IN == IN_BNRELU) {
        // s_aux[c] = scale (gamma * rstd), s_aux[kMax + c] = mean, s_aux[2 kMax + c] = beta.
        // z = (x - mean) * scale + beta: subtracting the mean FIRST keeps z accurate near zero, so the
        // ReLU mask (and hence every gradient) flips no more often than in the reference's fp32 path.
        for (int c = tid; c < p.cin; c += kConvThreads) {
            float scale, mean, beta;
            bn_input_constants(p, p0, grp, groups, first_of_group, c, scale, mean, beta);
            s_aux[c] = scale;
            s_aux[kMaxBnChannels + c] = mean;
            s_aux[2 * kMaxBnChannels + c] = beta;
        }
    }
    if constexpr (EPI == EPI_DGRAD_BN) {
        // per output channel of this block: scale, beta, mean, rstd at s_aux[3*kMax + 4*j ..]
        float* cst = s_aux + 3 * kMaxBnChannels;
        if (tid < NB) {
            const int c = co_base + tid;
            float mean = 0.f, rstd = 0.f, scale = 0.f, beta = 0.f;
            if (c < p.cout) {
                mean = p.bn_saved[2 * c];
                rstd = p.bn_saved[2 * c + 1];
                scale = p.bn_gamma[c] * rstd;
                beta = p.bn_beta[c];
            }
            cst[4 * tid] = scale; cst[4 * tid + 1] = beta; cst[4 * tid + 2] = mean; cst[4 * tid + 3] = rstd;
        }
    }

    // ---------------- accumulators ----------------
    f32x4 acc[R][Q];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wx = (wave % WX) * 16;     // wave's x offset inside the tile
    const int wy = (wave / WX) * R;      // wave's first row inside the tile
    const int li = lane & 15;
    const int lk = lane >> 4;

    // Each thread owns kPos fixed positions of the haloed tile and walks the chunk's channels over
    // them, so the address / bounds arithmetic is done once per block, not once per element.
    float pre[G::kPre];
    float wpre[kWPre];
    int goff[G::kPos];
    unsigned pos_ok = 0;
    unsigned pos_code = 0;
#pragma unroll
    for (int k = 0; k < G::kPos; ++k) {
        const int e = tid + k * kConvThreads;
        goff[k] = 0;
        if (e < G::kPlane) {
            const int ry = e / G::kCols;
            const int rx = e - ry * G::kCols;
            const int gy = y0 - G::kHalo + ry;
            const int gx = x0 - G::kHalo + rx;
            if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) {
                pos_ok |= (1u << k);
                if constexpr (IN == IN_UPSAMPLE || IN == IN_UNPOOL) {
                    goff[k] = (gy >> 1) * p.in_w + (gx >> 1);
                    pos_code |= static_cast<unsigned>(((gy & 1) << 1) | (gx & 1)) << (2 * k);
                } else {
                    goff[k] = gy * p.in_w + gx;
                }
            }
        }
    }
    const float* in_n = p.in + n * p.in_ns;
    const uint8_t* idx_n = nullptr;
    if constexpr (IN == IN_UNPOOL) idx_n = p.in_idx + n * p.idx_ns;
    const int nchunks = (p.cin + KC - 1) / KC;

    auto load_chunk = [&](int chunk) {
        const int c_base = chunk * KC;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int ch = c_base + c;
            const bool ch_ok = ch < p.cin;
            const int64_t coff = static_cast<int64_t>(ch) * p.in_cs;
#pragma unroll
            for (int k = 0; k < G::kPos; ++k) {
                float v = 0.f;
                if (ch_ok && (pos_ok & (1u << k))) {
                    if constexpr (IN == IN_UNPOOL) {
                        const int code = (pos_code >> (2 * k)) & 3;
                        v = (idx_n[coff + goff[k]] == code) ? in_n[coff + goff[k]] : 0.f;
                    } else {
                        v = in_n[coff + goff[k]];
                    }
                }
                pre[c * G::kPos + k] = v;
            }
        }
#pragma unroll
        for (int k = 0; k < kWPre; ++k) {
            const int e = tid + k * kConvThreads;
            float v = 0.f;
            if (e < kWElems) {
                const int j = e % NB;
                const int rest = e / NB;
                const int c = rest % KC;
                const int tap = rest / KC;
                const int ci = c_base + c;
                const int co = co_base + j;
                if (ci < p.cin && co < p.cout) {
                    if constexpr (kDgrad) {
                        // logical (ci, co, tap) = original (cout = ci, cin = co, flipped tap)
                        v = p.wgt[(static_cast<int64_t>(ci) * p.w_cin + co) * KK + (KK - 1 - tap)];
                    } else {
                        v = p.wgt[(static_cast<int64_t>(co) * p.w_cin + ci) * KK + tap];
                    }
                }
            }
            wpre[k] = v;
        }
    };

    auto store_chunk = [&](int chunk) {
        const int c_base = chunk * KC;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
#pragma unroll
            for (int k = 0; k < G::kPos; ++k) {
                const int e = tid + k * kConvThreads;
                if (e < G::kPlane) {
                    float v = pre[c * G::kPos + k];
                    if constexpr (IN == IN_BNRELU) {
                        const int ch = c_base + c;
                        if (ch < p.cin && (pos_ok & (1u << k))) {
                            v = fmaf(v - s_aux[kMaxBnChannels + ch], s_aux[ch], s_aux[2 * kMaxBnChannels + ch]);
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
        for (int k = 0; k < kWPre; ++k) {
            const int e = tid + k * kConvThreads;
            if (e < kWElems) s_w[e] = wpre[k];
        }
    };

    load_chunk(0);
    __syncthreads();   // s_aux ready
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk > 0) __syncthreads();      // everyone done reading the previous tile
        store_chunk(chunk);
        __syncthreads();
        if (chunk + 1 < nchunks) load_chunk(chunk + 1);

#pragma unroll
        for (int quad = 0; quad < KC / 4; ++quad) {
            const float* a_base = s_in + (quad * 4 + lk) * G::kCS + wy * G::kCols + wx + li;
            const float* b_base = s_w + (quad * 4 + lk) * NB + li;
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
                float a[R + KS - 1];
#pragma unroll
                for (int r = 0; r < R + KS - 1; ++r) a[r] = a_base[r * G::kCols + dx];
#pragma unroll
                for (int dy = 0; dy < KS; ++dy) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const float b = b_base[(dy * KS + dx) * KC * NB + q * 16];
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r + dy], b, acc[r][q], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---------------- epilogue ----------------
    conv_epilogue<Q, EPI, R>(p, acc, s_aux + 3 * kMaxBnChannels, s_aux + 3 * kMaxBnChannels + 4 * NB, x0, y0, wx, wy, co_base, n);
}

template <int KS, int KC, int Q, int WX, int R>
constexpr size_t conv_smem_bytes() {
    return sizeof(float) * (KC * ConvGeom<KS, KC, WX, R>::kCS + KS * KS * KC * 16 * Q + 3 * kMaxBnChannels + 4 * 16 * Q + 4 * 16 * Q * 2);
}

// p.tiles_x is filled in here from the tile shape of the chosen instantiation
template <int KS, int KC, int Q, int IN, int EPI, int WX, int R>
inline int launch_conv(ConvParams p, hipStream_t stream) {
    using G = ConvGeom<KS, KC, WX, R>;
    static_assert(G::kPre <= 64 && G::kPos <= 16, "staging registers");
    p.tiles_x = (p.w + G::kTileX - 1) / G::kTileX;
    const int tiles_y = (p.h + G::kTileY - 1) / G::kTileY;
    dim3 grid(p.tiles_x * tiles_y, (p.cout + 16 * Q - 1) / (16 * Q), p.n);
    constexpr size_t smem = conv_smem_bytes<KS, KC, Q, WX, R>();
    static bool configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& configured = configured_by_device[dev & 15];
    if (!configured && smem > 48 * 1024) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma_kernel<KS, KC, Q, IN, EPI, WX, R>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
        configured = true;
    }
    conv_mfma_kernel<KS, KC, Q, IN, EPI, WX, R><<<grid, kConvThreads, smem, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// Pick the tile shape from the image size: big tiles where there are enough of them to fill the
// chip, small tiles on the coarse levels (fewer wasted MFMAs on partial tiles, more blocks).
template <int KS, int KC, int Q, int IN, int EPI, int BIG_R = 8>
inline int launch_conv_auto(const ConvParams& p, hipStream_t stream) {
    const long tiles_big = static_cast<long>((p.w + 31) / 32) * ((p.h + 15) / 16) * p.n;
    if (tiles_big >= 512) return launch_conv<KS, KC, Q, IN, EPI, 2, BIG_R>(p, stream);
    const long tiles_mid = static_cast<long>((p.w + 15) / 16) * ((p.h + 15) / 16) * p.n;
    if (tiles_mid >= 384) return launch_conv<KS, KC, Q, IN, EPI, 1, 4>(p, stream);
    constexpr int KCS = (KS == 3 && KC == 8) ? 16 : KC;     // fewer, fatter K-chunks when a block is alone on its CU
    return launch_conv<KS, KCS, Q, IN, EPI, 1, 2>(p, stream);
}

}  // namespace endo
