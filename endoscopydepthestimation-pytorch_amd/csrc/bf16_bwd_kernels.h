// bf16-STORAGE kernel family, backward pass (round 3): what is not a convolution over the gradient buffer (those run through
// bf16_conv_kernel with its kEpiDgradBn / kEpiSumPool epilogues): the final 1 x 1 + |.| backward, the deferred BatchNorm terms
// (prep_dy + finalize, the scheme of the fp32 family, net.hip prep_dy_kernel / bn_bwd_finalize_kernel), and the weight gradients.
//
// Weight gradient of a convolution over 32-channel-blocked bf16 buffers.  dW[co][ci][ky][kx] = sum over pixels of
// G[co][y][x] * a[ci][y + ky - 1][x + kx - 1], a = relu(bn(x)) recomputed from the stored bf16 x (reference models.py:22-25).  On
// v_mfma_f32_16x16x32_bf16 the contraction index k is the PIXEL (8 consecutive pixels of one channel per lane), which the blocked
// layout keeps 64 bytes apart: both operands are transposed through LDS on the way in ([channel][row][32 pixels], channel pitch
// 16 bytes off a multiple of 256 so that the 16 channels of a fragment read land in 16 different bank groups).  A = G (i = cout), B = a
// (j = ci): the SMALL operand is the one shifted per tap -- G is staged three times, pre-shifted by kx - 1 pixels, so that every
// fragment read is 16-byte aligned; the ky shift is a row offset.  One B fragment feeds 9 MFMAs (3 x 3: the 9 taps of one 16-cout
// group; 1 x 1: 9 cout groups).  A block owns up to 192 input channels, 3 ci tiles per wave, 27 accumulator tiles per wave, and walks
// 4 x 32 pixel tiles; its sums leave as one fp32 partial per block, reduced by a second small kernel (deterministic, no atomics).
#pragma once

#include "bf16_conv_kernels.h"

namespace endo {
inline namespace ENDO16_NS {

__device__ __forceinline__ int rot_index(int c, int rot, int rot_n) { return c < rot_n ? (c + rot < rot_n ? c + rot : c + rot - rot_n) : c; }
// element offset of channel ca of pixel pix in a [t / blk][plane][blk] sample
__device__ __forceinline__ int64_t blk_off(int ca, int64_t pix, int64_t plane, int blk) {
    const int cb = ca / blk;
    return (cb * plane + pix) * blk + (ca - cb * blk);
}

// ---- gradient scale (half storage only): S = the power of two that brings max |grad_out| to ~2^9; every stored gradient carries the factor
// S, the parameter gradients are multiplied by 1 / S where they leave (weight-gradient reduction, BatchNorm parameters, biases); the
// deferred BatchNorm terms are linear in the gradient and carry S by themselves.  `scale` = {S, 1 / S, max bits, -}: the caller zeroes
// word 2, s16_grad_max_kernel's blocks fold their maxima into it (non-negative floats order like their bit patterns) and the one-thread
// s16_grad_scale_kernel that follows writes S.  (Two launches rather than "the last block finalises": that needs a device-scope fence
// per block, which on this part writes back and invalidates an XCD's L2.)  Head-room assumption: 2^9 at the output
// leaves a factor 2^6.9 up to half's largest finite value for the growth of per-pixel gradients inside the network; the parity tests
// run with 1e-6-scaled and O(1) output gradients, and a saturated store shows up as a non-finite parameter gradient.
__global__ void __launch_bounds__(1024) s16_grad_max_kernel(const float* __restrict__ g, int64_t count, float* __restrict__ scale) {
    __shared__ float s_max[16];
    float m = 0.f;
    for (int64_t i = blockIdx.x * 1024ll + threadIdx.x; i < count; i += 1024ll * gridDim.x) m = fmaxf(m, fabsf(g[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = fmaxf(m, s_max[i]);
        atomicMax(reinterpret_cast<unsigned*>(scale) + 2, __float_as_uint(m));
    }
}

__global__ void s16_grad_scale_kernel(float* __restrict__ scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float m = __uint_as_float(reinterpret_cast<const unsigned*>(scale)[2]);
    float sc = 1.f;
    if (m > 0.f && m < 3.0e38f) {
        int e = 9 - static_cast<int>(floorf(log2f(m)));
        e = e < -24 ? -24 : (e > 40 ? 40 : e);
        sc = exp2f(static_cast<float>(e));
    }
    scale[0] = sc; scale[1] = 1.f / sc;
}

// ---- final 1 x 1 + |.| backward (reference models.py:167, 186): du[c] = g * sign(pre) * w[c] for all 192 channels (first writer of the
// level-0 gradient buffer); grad_w[c] += sum g * sign(pre) * u[c]; grad_b += sum g * sign(pre).  A lane owns one 16-byte unit (8 channels)
// of a pixel record and walks the 6 channel blocks: a wave's access is 16 pixels x 64 bytes = 1 KiB contiguous per block, read and write.
// (8 lanes per pixel with 24 channels each -- three 16-byte pieces in different blocks per lane -- ran at 2.4 TB/s.)
__global__ void __launch_bounds__(256) bf16_final_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ pre,
                                                             const uint16_t* __restrict__ u, uint16_t* __restrict__ du, int64_t ns, int plane,
                                                             const float* __restrict__ w, int rot, int rot_n, float* __restrict__ grad_w,
                                                             float* __restrict__ grad_b, double* __restrict__ gsum, const float* __restrict__ gscale) {
    constexpr int kBlocks = 192 / 32;
    __shared__ float s_part[4][4][kBlocks * 8 + 1];
    const int n = blockIdx.y;
    const int part = threadIdx.x & 3;
    float wv[kBlocks][8], gw[kBlocks][8];
    float gb = 0.f;
    const float S = gscale ? gscale[0] : 1.f;          // what is written to the gradient buffer carries S; the parameter gradients here do not
#pragma unroll
    for (int cb = 0; cb < kBlocks; ++cb)
#pragma unroll
        for (int k = 0; k < 8; ++k) { wv[cb][k] = S * w[rot_index(cb * 32 + part * 8 + k, rot, rot_n)]; gw[cb][k] = 0.f; }
    for (int px = (blockIdx.x * blockDim.x + threadIdx.x) >> 2; px < plane; px += (gridDim.x * blockDim.x) >> 2) {
        const float z = pre[static_cast<int64_t>(n) * plane + px];
        const float gs = gout[static_cast<int64_t>(n) * plane + px] * (z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f));
        if (part == 0) gb += gs;
        u32x4_t v[kBlocks];
#pragma unroll
        for (int cb = 0; cb < kBlocks; ++cb)
            v[cb] = *reinterpret_cast<const u32x4_t*>(u + n * ns + (static_cast<int64_t>(cb) * plane + px) * 32 + part * 8);
#pragma unroll
        for (int cb = 0; cb < kBlocks; ++cb) {
            u32x4_t o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                gw[cb][2 * k] = fmaf(gs, s16_lo(v[cb][k]), gw[cb][2 * k]);
                gw[cb][2 * k + 1] = fmaf(gs, s16_hi(v[cb][k]), gw[cb][2 * k + 1]);
                o[k] = pack_s16x2(gs * wv[cb][2 * k], gs * wv[cb][2 * k + 1]);
            }
            *reinterpret_cast<u32x4_t*>(du + n * ns + (static_cast<int64_t>(cb) * plane + px) * 32 + part * 8) = o;
        }
    }
    // lanes with the same part: 16 per wave (lane bits 2..5), then the 4 waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int cb = 0; cb < kBlocks; ++cb)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = gw[cb][k];
            t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
            gw[cb][k] = t;
        }
    gb += __shfl_xor(gb, 4, 64); gb += __shfl_xor(gb, 8, 64); gb += __shfl_xor(gb, 16, 64); gb += __shfl_xor(gb, 32, 64);
    if (lane < 4) {
#pragma unroll
        for (int cb = 0; cb < kBlocks; ++cb)
#pragma unroll
            for (int k = 0; k < 8; ++k) s_part[wave][part][cb * 8 + k] = gw[cb][k];
        s_part[wave][part][kBlocks * 8] = gb;
    }
    __syncthreads();
    const float gb_block = s_part[0][0][kBlocks * 8] + s_part[1][0][kBlocks * 8] + s_part[2][0][kBlocks * 8] + s_part[3][0][kBlocks * 8];
    if (threadIdx.x < 192) {
        const int c = threadIdx.x, e = (c >> 5) * 8 + (c & 7), pt = (c & 31) >> 3;
        atomicAdd(grad_w + rot_index(c, rot, rot_n), s_part[0][pt][e] + s_part[1][pt][e] + s_part[2][pt][e] + s_part[3][pt][e]);
        // sum over pixels of what this kernel writes into channel c of the gradient buffer (see bf16_prep_dy_kernel)
        atomicAdd(gsum + 2 * c, static_cast<double>(w[rot_index(c, rot, rot_n)]) * static_cast<double>(gb_block) * static_cast<double>(S));
    }
    if (threadIdx.x == 192) atomicAdd(grad_b, gb_block);
}

// ---- prep_dy: a channel range [c0, c0 + count) of a level's gradient buffer becomes the full gradient of the stored values once
// all its consumers have been differentiated: d += P[c] * x + Q[c] (the deferred BatchNorm terms, zero in inference mode).  count % 4 == 0.
//
// The bias gradient of the convolution that produced the range (reference: conv bias, models.py:24) is the sum of that full gradient
// over the pixels -- and is NOT taken from the bf16 buffer: d holds sums of scale * da that the deferred terms then largely cancel
// (BatchNorm removes the mean of the gradient), so its 8-bit roundings are relative to the uncancelled magnitude and their sum over
// 1e4..1e6 pixels was measured at 2-3x the true bias gradient.  Every contribution's pixel sum is known exactly where it is made:
// a training-mode BatchNorm consumer contributes 0 (scale * S1 - scale * S1 - k * sum(x - mean)), an inference-mode one scale * S1
// (bf16_bn_finalize_kernel), the final convolution w[c] * sum g (bf16_final_bwd_kernel), a transition up the fp32 sum of what its
// kEpiSumPool epilogue adds.  They meet in gsum ([t][2] fp64 per level, first of each pair), read here by one block.
__global__ void __launch_bounds__(256) bf16_prep_dy_kernel(uint16_t* __restrict__ d, const uint16_t* __restrict__ x, int64_t ns, int plane, int blk,
                                                           int c0, int count, const float* __restrict__ pq_p, const float* __restrict__ pq_q,
                                                           float* __restrict__ bias_grad, const double* __restrict__ gsum, int apply, int group_n,
                                                           int64_t gs_pq, const float* __restrict__ gscale) {
    if (bias_grad && blockIdx.x == 0 && blockIdx.y == 0)
        for (int c = threadIdx.x; c < count; c += 256) bias_grad[c] += static_cast<float>(gsum[2 * (c0 + c)] * (gscale ? gscale[1] : 1.f));
    if (!apply) return;          // inference mode: P = Q = 0
    const int n = blockIdx.y;
    const int quads = count >> 2;
    const int ppi = 256 / quads;                                   // pixels per iteration of the block
    const int q = threadIdx.x % quads, p0 = threadIdx.x / quads;
    if (p0 >= ppi) return;
    float pc[4], qc[4];
    const int64_t go = (group_n > 0 ? n / group_n : 0) * gs_pq;          // the sample's group has its own deferred terms
#pragma unroll
    for (int i = 0; i < 4; ++i) { pc[i] = pq_p[go + c0 + 4 * q + i]; qc[i] = pq_q[go + c0 + 4 * q + i]; }
    for (int px = blockIdx.x * ppi + p0; px < plane; px += gridDim.x * ppi) {
        const int64_t off = n * ns + blk_off(c0 + 4 * q, px, plane, blk);
        const u32x2_t dv = *reinterpret_cast<const u32x2_t*>(d + off);
        const u32x2_t xv = *reinterpret_cast<const u32x2_t*>(x + off);
        const float g0 = s16_lo(dv[0]) + fmaf(pc[0], s16_lo(xv[0]), qc[0]), g1 = s16_hi(dv[0]) + fmaf(pc[1], s16_hi(xv[0]), qc[1]);
        const float g2 = s16_lo(dv[1]) + fmaf(pc[2], s16_lo(xv[1]), qc[2]), g3 = s16_hi(dv[1]) + fmaf(pc[3], s16_hi(xv[1]), qc[3]);
        // stochastic rounding: P x + Q is mostly below half an ulp of d (pack_s16x2_sr)
        const unsigned key = (static_cast<unsigned>(off - n * ns) + static_cast<unsigned>(group_n > 0 ? n % group_n : n) * 0x632BE5ABu) ^ 0x51ED270Bu;
        *reinterpret_cast<u32x2_t*>(d + off) = u32x2_t{pack_s16x2_sr(g0, g1, key), pack_s16x2_sr(g2, g3, key + 2)};
    }
}

// ---- BatchNorm backward, the per-channel part (reference: nn.BatchNorm2d in front of every dense / transition-down convolution).
// sums: [cin][2] fp64 from the kEpiDgradBn epilogue (S1 = sum da, S2raw = sum da * x); xhat = (x - mean) * rstd, so
// S2 = sum da * xhat = rstd * (S2raw - mean * S1).  ggamma += S2, gbeta += S1; training mode also defers
//   dx = scale * da - scale * S1 / M - scale * rstd * (S2 / M) * (x - mean)   =>   P += -scale * rstd * S2 / M,
//   Q += -scale * S1 / M + scale * rstd * S2 / M * mean        (the fp32 family's bn_bwd_finalize_kernel, net.hip)
// to the prep_dy of the channels (buffer channel ic0 + ci; parameters at (ci + rot) % rot_n).
__global__ void __launch_bounds__(128) bf16_bn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ saved,
                                                               const float* __restrict__ gamma, float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                               float* __restrict__ pq_p, float* __restrict__ pq_q, double* __restrict__ gsum, int first,
                                                               int cnt, int rot, int rot_n, double count, int training, int64_t gs_sums, int64_t gs_saved,
                                                               int64_t gs_pq, const float* __restrict__ gscale, int64_t slot_stride) {
    const double inv = gscale ? gscale[1] : 1.0;          // the parameter gradients leave without the gradient scale
    // blockIdx.y = sample group: its own sums, statistics and deferred terms; the parameter gradients add up over the groups
    sums += blockIdx.y * gs_sums; saved += blockIdx.y * gs_saved; pq_p += blockIdx.y * gs_pq; pq_q += blockIdx.y * gs_pq;
    for (int ci = first + blockIdx.x * blockDim.x + threadIdx.x; ci < first + cnt; ci += gridDim.x * blockDim.x) {
        const int pc = rot_index(ci, rot, rot_n);
        const double mean = saved[2 * pc], rstd = saved[2 * pc + 1];
        const double s1 = bn_slot_sum(sums + 2 * ci, slot_stride), s2 = rstd * (bn_slot_sum(sums + 2 * ci + 1, slot_stride) - mean * s1);
        atomicAdd(ggamma + pc, static_cast<float>(s2 * inv));
        atomicAdd(gbeta + pc, static_cast<float>(s1 * inv));
        const double scale = gamma[pc] * rstd;
        if (training) {
            const double k = scale * rstd * s2 / count;
            pq_p[ci] += static_cast<float>(-k);
            pq_q[ci] += static_cast<float>(-scale * s1 / count + k * mean);
        } else {
            atomicAdd(gsum + 2 * ci, scale * s1);          // the pixel sum of what the layer added to channel ci (training mode: exactly 0)
        }
    }
}

// the same for the four BN layers of a dense block over the block's base channels [0, cnt): one launch instead of four, the layers' terms added
// to P and Q in the order (and with the fp32 roundings) of four single-layer launches
struct BnFin16x4 {
    const double* sums[4];
    const float* saved[4];
    const float* gamma[4];
    float* ggamma[4];
    float* gbeta[4];
    int rot[4], rot_n[4];
};
__global__ void __launch_bounds__(128) bf16_bn_finalize4_kernel(const BnFin16x4 a, float* __restrict__ pq_p, float* __restrict__ pq_q, double* __restrict__ gsum,
                                                                int cnt, double count, int training, int64_t gs_sums, int64_t gs_saved, int64_t gs_pq,
                                                                const float* __restrict__ gscale, int64_t slot_stride) {
    const double inv = gscale ? gscale[1] : 1.0;
    pq_p += blockIdx.y * gs_pq; pq_q += blockIdx.y * gs_pq;
    for (int ci = blockIdx.x * blockDim.x + threadIdx.x; ci < cnt; ci += gridDim.x * blockDim.x) {
        float pp = pq_p[ci], qq = pq_q[ci];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const double* sums = a.sums[l] + blockIdx.y * gs_sums;
            const float* saved = a.saved[l] + blockIdx.y * gs_saved;
            const int pc = rot_index(ci, a.rot[l], a.rot_n[l]);
            const double mean = saved[2 * pc], rstd = saved[2 * pc + 1];
            const double s1 = bn_slot_sum(sums + 2 * ci, slot_stride), s2 = rstd * (bn_slot_sum(sums + 2 * ci + 1, slot_stride) - mean * s1);
            atomicAdd(a.ggamma[l] + pc, static_cast<float>(s2 * inv));
            atomicAdd(a.gbeta[l] + pc, static_cast<float>(s1 * inv));
            const double scale = a.gamma[l][pc] * rstd;
            if (training) {
                const double k = scale * rstd * s2 / count;
                pp += static_cast<float>(-k);
                qq += static_cast<float>(-scale * s1 / count + k * mean);
            } else {
                atomicAdd(gsum + 2 * ci, scale * s1);
            }
        }
        if (training) { pq_p[ci] = pp; pq_q[ci] = qq; }
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------
constexpr int kWgRows = 4, kWgCols = 32;                 // pixels of a tile
// input channels per block = 64 T (grid.y covers the rest): T 16-channel tiles per wave.  3 x 3: T = 1 -- 9 accumulator tiles per wave instead
// of 27 and a third of the LDS; a tile's activations are then ONE batch of four 16-byte units per thread, and the registers this frees hold
// the NEXT tile's loads (G window and activation batch) across the matrix phase: the walk over the tiles no longer waits for memory twice
// per tile with nothing else to do
constexpr int kWgT3 = 1, kWgT1 = 3;
constexpr int bf16_wgrad_tiles_per_wave(int ks) { return ks == 3 ? kWgT3 : kWgT1; }
constexpr int kWgPitchA = kWgRows * 64 + 16;             // bytes between channels of the staged a tile
constexpr int kWgPitchG3 = (kWgRows + 2) * 3 * 64 + 16;  // 3 x 3: [16 cout][6 rows][3 shifted copies][32 px]
constexpr int kWgCo1 = 144;                              // 1 x 1: couts per block (9 fragments of 16)

struct Wgrad16Params {
    int n, h, w;                       // grid of G (the forward convolution's output)
    const uint16_t* a;                 // forward input of the convolution (bf16, blocked)
    int64_t a_ns;
    int a_blk, a_h, a_w;               // a_h, a_w = h, w unless ups
    int ac0, cin;                      // the convolution read channels [ac0, ac0 + cin)
    int cin_w;                         // input channels of the weight tensor (0 = cin; the first convolution's 3 travel as 4)
    int ups;                           // nearest x2 upsampling of a (transition up)
    const float* saved;                // BatchNorm of the input: (mean, rstd) at parameter index (ci + rot) % rot_n, or null = raw input
    const float* gamma;
    const float* beta;
    int rot, rot_n;
    int group_n;                       // sample groups (at most 2): group g's saved starts gs_saved floats later
    int64_t gs_saved;
    const uint16_t* g;                 // prepared gradient of the convolution's output
    int64_t g_ns;
    int g_blk;
    int gc0, cout;
    const uint8_t* g_idx;              // 1 x 1 only: g is the POOLED gradient ([h / 2][w / 2]) and g_idx the forward max-pool codes
    float* partial;                    // [gridDim.x][co groups][9][16][ci_pad] fp32
    int64_t partial_cap;               // floats the caller reserved behind `partial` (0 = unchecked: stand-alone benches)
    const float* gscale;               // {S, 1 / S} of the stored gradients or null (s16_grad_scale_kernel)
    int ci_pad;                        // cin rounded up to 16
};

// KS = 3: grid (blocks, ci groups of 192, cout groups of 16); KS = 1: grid (blocks, ci groups of 192, cout groups of 144)
template <int KS, int T>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) bf16_wgrad_kernel(const Wgrad16Params p) {
    constexpr int kWgCi = 64 * T;
    constexpr int kGBytes = KS == 3 ? 16 * kWgPitchG3 : kWgCo1 * kWgPitchA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wg[];
    unsigned char* s_a = smem_wg;                                       // [64 T][4 rows][32 px] bf16, pitch kWgPitchA
    unsigned char* s_g = s_a + kWgCi * kWgPitchA;
    float* s_bn = reinterpret_cast<float*>(s_g + kGBytes);              // [2 groups][192][2] (scale, shift)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int ci0 = blockIdx.y * kWgCi;
    const int cin_g = p.cin - ci0 < kWgCi ? p.cin - ci0 : kWgCi;       // this block's input channels
    const int units_c = (cin_g + 7) >> 3;                               // 8-channel units per pixel
    const int ntiles_ci = (cin_g + 15) >> 4;
    const int co0 = blockIdx.z * (KS == 3 ? 16 : kWgCo1);
    const int cout_g = p.cout - co0 < (KS == 3 ? 16 : kWgCo1) ? p.cout - co0 : (KS == 3 ? 16 : kWgCo1);
    const int a_plane = p.a_h * p.a_w;
    const int g_h = p.g_idx ? p.h >> 1 : p.h, g_w = p.g_idx ? p.w >> 1 : p.w;
    const int g_plane = g_h * g_w;

    for (int e = tid; e < 2 * kWgCi; e += 256) {
        const int g = e / kWgCi, c = e - g * kWgCi;
        float sc = 1.f, sh = 0.f;
        if (p.saved && c < cin_g && (g == 0 || p.group_n > 0)) {
            const int pc = rot_index(ci0 + c, p.rot, p.rot_n);
            const float* sv = p.saved + g * p.gs_saved;
            sc = p.gamma[pc] * sv[2 * pc + 1];
            sh = fmaf(-sv[2 * pc], sc, p.beta[pc]);
        }
        s_bn[2 * e] = sc; s_bn[2 * e + 1] = sh;
    }

    f32x4_t acc[9][T];
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[f][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int tiles_x = (p.w + kWgCols - 1) / kWgCols, tiles_y = (p.h + kWgRows - 1) / kWgRows;
    const int total = tiles_x * tiles_y * p.n;
    if constexpr (KS == 3 && T == 1) {
        // ---- pipelined walk: tile i + 1's G window and activation batch are loaded while tile i is in its matrix phase ----
        constexpr int kGItems = (kWgRows + 2) * (kWgCols + 2) * 4, kGIter = (kGItems + 255) / 256;
        const int quads = (cout_g + 3) >> 2;
        u32x2_t gv[kGIter];
        u32x4_t av[4];
        unsigned aok = 0;
        auto tile_geom = [&](int tile, int& n, int& y0, int& x0) {
            n = tile / (tiles_x * tiles_y);
            const int rem = tile - n * tiles_x * tiles_y;
            y0 = (rem / tiles_x) * kWgRows; x0 = (rem % tiles_x) * kWgCols;
        };
        auto issue = [&](int tile) {
            int n, y0, x0;
            tile_geom(tile, n, y0, x0);
            const bool live = tile < total;
#pragma unroll
            for (int i = 0; i < kGIter; ++i) {
                const int it = tid + i * 256;
                const int q = it % quads, px = it / quads;
                const int ry = px / (kWgCols + 2), rx = px - ry * (kWgCols + 2);          // tile pixel (ry - 1, rx - 1)
                const int gy = y0 + ry - 1, gx = x0 + rx - 1;
                gv[i] = u32x2_t{0u, 0u};
                if (live && it < (kWgRows + 2) * (kWgCols + 2) * quads && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w)
                    gv[i] = *reinterpret_cast<const u32x2_t*>(p.g + n * p.g_ns + blk_off(p.gc0 + co0 + 4 * q, static_cast<int64_t>(gy) * p.w + gx, g_plane, p.g_blk));
            }
            aok = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {          // unit k = (it >> 9) * 4 + (it & 3) of pixel (it >> 2) & 127, it = i * 256 + tid: 8 units of 8 channels
                const int it = i * 256 + tid;
                const int k = (it >> 9) * 4 + (it & 3), px = (it >> 2) & 127;
                const int gy = y0 + (px >> 5), gx = x0 + (px & 31);
                av[i] = u32x4_t{0u, 0u, 0u, 0u};
                if (live && k < units_c && gx < p.w && gy < p.h) {
                    aok |= 1u << i;
                    const int sy = p.ups ? gy >> 1 : gy, sx = p.ups ? gx >> 1 : gx;
                    av[i] = *reinterpret_cast<const u32x4_t*>(p.a + n * p.a_ns + blk_off(p.ac0 + ci0 + 8 * k, static_cast<int64_t>(sy) * p.a_w + sx, a_plane, p.a_blk));
                }
            }
        };
        issue(blockIdx.x);
        for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
            const int n = tile / (tiles_x * tiles_y);
            const float* bn_g = s_bn + (p.group_n > 0 ? n / p.group_n : 0) * 2 * kWgCi;
            __syncthreads();          // the previous tile's fragment reads (and, the first time, s_bn)
#pragma unroll
            for (int i = 0; i < kGIter; ++i) {
                const int it = tid + i * 256;
                if (it >= (kWgRows + 2) * (kWgCols + 2) * quads) continue;
                const int q = it % quads, px = it / quads;
                const int ry = px / (kWgCols + 2), rx = px - ry * (kWgCols + 2);
                const u32x2_t v = gv[i];
                const uint16_t e[4] = {static_cast<uint16_t>(v[0] & 0xffffu), static_cast<uint16_t>(v[0] >> 16), static_cast<uint16_t>(v[1] & 0xffffu),
                                       static_cast<uint16_t>(v[1] >> 16)};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int x = rx - 1 + kx - 1;          // copy kx holds G[x - kx + 1] at x
                    if (x < 0 || x >= kWgCols) continue;
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2)
                        *reinterpret_cast<uint16_t*>(s_g + (4 * q + i2) * kWgPitchG3 + (ry * 3 + kx) * 64 + x * 2) = e[i2];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int it = i * 256 + tid;
                const int k = (it >> 9) * 4 + (it & 3), px = (it >> 2) & 127;
                if (k >= units_c) continue;
                const int ry = px >> 5, rx = px & 31;
                const u32x4_t v = av[i];
                const bool ok = (aok >> i) & 1u;
                float z[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { z[2 * j] = s16_lo(v[j]); z[2 * j + 1] = s16_hi(v[j]); }
                if (p.saved) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4_t q = *reinterpret_cast<const f32x4_t*>(bn_g + 2 * (8 * k + 2 * j));
                        z[2 * j] = ok ? fmaxf(fmaf(z[2 * j], q[0], q[1]), 0.f) : 0.f;
                        z[2 * j + 1] = ok ? fmaxf(fmaf(z[2 * j + 1], q[2], q[3]), 0.f) : 0.f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned pk = pack_s16x2(z[2 * j], z[2 * j + 1]);
                    *reinterpret_cast<uint16_t*>(s_a + (8 * k + 2 * j) * kWgPitchA + ry * 64 + rx * 2) = static_cast<uint16_t>(pk & 0xffffu);
                    *reinterpret_cast<uint16_t*>(s_a + (8 * k + 2 * j + 1) * kWgPitchA + ry * 64 + rx * 2) = static_cast<uint16_t>(pk >> 16);
                }
            }
            issue(tile + gridDim.x);          // flies under the barrier and the matrix phase
            __syncthreads();
            if (wave < ntiles_ci) {
#pragma unroll
                for (int row = 0; row < kWgRows; ++row) {
                    const s16x8_t b = *reinterpret_cast<const s16x8_t*>(s_a + (16 * wave + li) * kWgPitchA + row * 64 + lk * 16);
#pragma unroll
                    for (int f = 0; f < 9; ++f) {
                        const s16x8_t a = *reinterpret_cast<const s16x8_t*>(s_g + li * kWgPitchG3 + ((row - f / 3 + 2) * 3 + f % 3) * 64 + lk * 16);
                        acc[f][0] = S16_MFMA(a, b, acc[f][0], 0, 0, 0);
                    }
                }
            }
        }
    } else
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int n = tile / (tiles_x * tiles_y);
        const int rem = tile - n * tiles_x * tiles_y;
        const int y0 = (rem / tiles_x) * kWgRows, x0 = (rem % tiles_x) * kWgCols;
        const float* bn_g = s_bn + (p.group_n > 0 ? n / p.group_n : 0) * 2 * kWgCi;
        __syncthreads();          // the previous tile's fragment reads (and, the first time, s_bn)
        // ---- G ----
        if constexpr (KS == 3) {
            const int quads = (cout_g + 3) >> 2;
            // (loads of a batch first, then its LDS writes: a load -> store loop pays the memory latency once per item)
            constexpr int kGItems = (kWgRows + 2) * (kWgCols + 2) * 4, kGIter = (kGItems + 255) / 256;
            u32x2_t gv[kGIter];
#pragma unroll
            for (int i = 0; i < kGIter; ++i) {
                const int it = tid + i * 256;
                const int q = it % quads, px = it / quads;
                const int ry = px / (kWgCols + 2), rx = px - ry * (kWgCols + 2);          // tile pixel (ry - 1, rx - 1)
                const int gy = y0 + ry - 1, gx = x0 + rx - 1;
                gv[i] = u32x2_t{0u, 0u};
                if (it < (kWgRows + 2) * (kWgCols + 2) * quads && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w)
                    gv[i] = *reinterpret_cast<const u32x2_t*>(p.g + n * p.g_ns + blk_off(p.gc0 + co0 + 4 * q, static_cast<int64_t>(gy) * p.w + gx, g_plane, p.g_blk));
            }
#pragma unroll
            for (int i = 0; i < kGIter; ++i) {
                const int it = tid + i * 256;
                if (it >= (kWgRows + 2) * (kWgCols + 2) * quads) continue;
                const int q = it % quads, px = it / quads;
                const int ry = px / (kWgCols + 2), rx = px - ry * (kWgCols + 2);
                const u32x2_t v = gv[i];
                const uint16_t e[4] = {static_cast<uint16_t>(v[0] & 0xffffu), static_cast<uint16_t>(v[0] >> 16), static_cast<uint16_t>(v[1] & 0xffffu),
                                       static_cast<uint16_t>(v[1] >> 16)};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int x = rx - 1 + kx - 1;          // copy kx holds G[x - kx + 1] at x
                    if (x < 0 || x >= kWgCols) continue;
#pragma unroll
                    for (int i2 = 0; i2 < 4; ++i2)
                        *reinterpret_cast<uint16_t*>(s_g + (4 * q + i2) * kWgPitchG3 + (ry * 3 + kx) * 64 + x * 2) = e[i2];
                }
            }
        } else {
            const int units_g = (cout_g + 7) >> 3;
            for (int it = tid; it < kWgRows * kWgCols * units_g; it += 256) {
                const int k = it % units_g, px = it / units_g;
                const int ry = px >> 5, rx = px & 31;
                const int gy = y0 + ry, gx = x0 + rx;
                u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
                if (gx < p.w && gy < p.h) {
                    if (p.g_idx) {
                        const int64_t pp = static_cast<int64_t>(gy >> 1) * g_w + (gx >> 1);
                        v = *reinterpret_cast<const u32x4_t*>(p.g + n * p.g_ns + blk_off(p.gc0 + co0 + 8 * k, pp, g_plane, p.g_blk));
                        const u32x2_t cw = *reinterpret_cast<const u32x2_t*>(p.g_idx + (static_cast<int64_t>(n) * g_plane + pp) * p.cout + co0 + 8 * k);
                        const unsigned pos = ((gy & 1) << 1) | (gx & 1);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const unsigned c2 = cw[j >> 1] >> (16 * (j & 1));
                            v[j] &= (((c2 & 0xffu) == pos) ? 0x0000ffffu : 0u) | ((((c2 >> 8) & 0xffu) == pos) ? 0xffff0000u : 0u);
                        }
                    } else {
                        v = *reinterpret_cast<const u32x4_t*>(p.g + n * p.g_ns + blk_off(p.gc0 + co0 + 8 * k, static_cast<int64_t>(gy) * p.w + gx, g_plane, p.g_blk));
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    *reinterpret_cast<uint16_t*>(s_g + (8 * k + 2 * j) * kWgPitchA + ry * 64 + rx * 2) = static_cast<uint16_t>(v[j] & 0xffffu);
                    *reinterpret_cast<uint16_t*>(s_g + (8 * k + 2 * j + 1) * kWgPitchA + ry * 64 + rx * 2) = static_cast<uint16_t>(v[j] >> 16);
                }
            }
        }
        // ---- a: relu(bn(x)) of the tile's own pixels, transposed to [channel][row][px]; u = ((k_hi * 128 + px) * 4 + k_lo), unit k_hi * 4 + k_lo ----
        const int units_hi = (units_c + 3) >> 2;
        for (int it0 = tid; it0 < units_hi * 512; it0 += 4 * 256) {          // batches of 4 units per thread: loads, then the rest
            u32x4_t av[4];
            bool aok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int it = it0 + i * 256;
                const int k = (it >> 9) * 4 + (it & 3), px = (it >> 2) & 127;
                const int gy = y0 + (px >> 5), gx = x0 + (px & 31);
                av[i] = u32x4_t{0u, 0u, 0u, 0u};
                aok[i] = it < units_hi * 512 && k < units_c && gx < p.w && gy < p.h;
                if (aok[i]) {
                    const int sy = p.ups ? gy >> 1 : gy, sx = p.ups ? gx >> 1 : gx;
                    av[i] = *reinterpret_cast<const u32x4_t*>(p.a + n * p.a_ns + blk_off(p.ac0 + ci0 + 8 * k, static_cast<int64_t>(sy) * p.a_w + sx, a_plane, p.a_blk));
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int it = it0 + i * 256;
                const int k = (it >> 9) * 4 + (it & 3), px = (it >> 2) & 127;
                if (it >= units_hi * 512 || k >= units_c) continue;
                const int ry = px >> 5, rx = px & 31;
                const u32x4_t v = av[i];
                const bool ok = aok[i];
                float z[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { z[2 * j] = s16_lo(v[j]); z[2 * j + 1] = s16_hi(v[j]); }
                if (p.saved) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4_t q = *reinterpret_cast<const f32x4_t*>(bn_g + 2 * (8 * k + 2 * j));
                        z[2 * j] = ok ? fmaxf(fmaf(z[2 * j], q[0], q[1]), 0.f) : 0.f;
                        z[2 * j + 1] = ok ? fmaxf(fmaf(z[2 * j + 1], q[2], q[3]), 0.f) : 0.f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned pk = pack_s16x2(z[2 * j], z[2 * j + 1]);
                    *reinterpret_cast<uint16_t*>(s_a + (8 * k + 2 * j) * kWgPitchA + ry * 64 + rx * 2) = static_cast<uint16_t>(pk & 0xffffu);
                    *reinterpret_cast<uint16_t*>(s_a + (8 * k + 2 * j + 1) * kWgPitchA + ry * 64 + rx * 2) = static_cast<uint16_t>(pk >> 16);
                }
            }
        }
        __syncthreads();
        // ---- this wave's ci tiles 3 wave .. 3 wave + 2, one k-step per tile row ----
        if (T * wave < ntiles_ci) {
#pragma unroll
            for (int row = 0; row < kWgRows; ++row) {
                s16x8_t b[T];
#pragma unroll
                for (int t = 0; t < T; ++t) b[t] = *reinterpret_cast<const s16x8_t*>(s_a + (16 * (T * wave + t) + li) * kWgPitchA + row * 64 + lk * 16);
#pragma unroll
                for (int f = 0; f < 9; ++f) {
                    s16x8_t a;
                    if constexpr (KS == 3) a = *reinterpret_cast<const s16x8_t*>(s_g + li * kWgPitchG3 + ((row - f / 3 + 2) * 3 + f % 3) * 64 + lk * 16);
                    else a = *reinterpret_cast<const s16x8_t*>(s_g + (16 * f + li) * kWgPitchA + row * 64 + lk * 16);
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[f][t] = S16_MFMA(a, b[t], acc[f][t], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial sums of the block: [block][cout group][f][16][ci_pad], lane = (ci = li of its tile, 4 couts 4 lk ..) ----
    float* dst = p.partial + ((static_cast<int64_t>(blockIdx.x) * gridDim.z + blockIdx.z) * 9) * 16 * p.ci_pad;
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int ci = ci0 + 16 * (T * wave + t) + li;
            if (T * wave + t >= ntiles_ci || ci >= p.ci_pad) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[(static_cast<int64_t>(f) * 16 + 4 * lk + i) * p.ci_pad + ci] = acc[f][t][i];
        }
}

// dW[co][(ci + rot) % rot_n][tap] += sum over blocks of partial; KS = 3: co = 16 z + r, tap = f; KS = 1: co = 144 z + 16 f + r.
// grid.y splits the blocks into slices of 64 (one fp32 atomic per element and slice): a single thread walking 512 partials is a chain
// of 512 dependent loads
__global__ void __launch_bounds__(256) bf16_wgrad_reduce_kernel(const float* __restrict__ partial, int blocks, int co_groups, int ci_pad, int cin, int cout,
                                                                int ks, int rot, int rot_n, float* __restrict__ dw, int cin_w, const float* __restrict__ gscale) {
    const float inv = gscale ? gscale[1] : 1.f;
    const int64_t per_block = static_cast<int64_t>(co_groups) * 9 * 16 * ci_pad;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < per_block; e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int ci = e % ci_pad;
        int64_t rest = e / ci_pad;
        const int r = rest & 15; rest >>= 4;
        const int f = rest % 9, z = rest / 9;
        const int co = ks == 3 ? 16 * z + r : kWgCo1 * z + 16 * f + r;
        if (ci >= cin_w || co >= cout) continue;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = blockIdx.y * 64;
        const int b_end = b + 64 < blocks ? b + 64 : blocks;
        for (; b + 3 < b_end; b += 4) {
            s0 += partial[b * per_block + e]; s1 += partial[(b + 1) * per_block + e];
            s2 += partial[(b + 2) * per_block + e]; s3 += partial[(b + 3) * per_block + e];
        }
        for (; b < b_end; ++b) s0 += partial[b * per_block + e];
        const int pci = rot_index(ci, rot, rot_n);
        atomicAdd(ks == 3 ? dw + (static_cast<int64_t>(co) * cin_w + pci) * 9 + f : dw + static_cast<int64_t>(co) * cin_w + pci, ((s0 + s1) + (s2 + s3)) * inv);
    }
}

template <int KS>
inline size_t bf16_wgrad_smem() {
    constexpr int kWgCi = 64 * bf16_wgrad_tiles_per_wave(KS);
    return static_cast<size_t>(kWgCi) * kWgPitchA + (KS == 3 ? 16 * kWgPitchG3 : kWgCo1 * kWgPitchA) + sizeof(float) * 2 * 2 * kWgCi;
}

// blocks of the walk over tiles, the partial buffer a launch needs (floats) and the launch itself
inline int bf16_wgrad_blocks(const Wgrad16Params& p, int ks) {
    const int tiles = ((p.w + kWgCols - 1) / kWgCols) * ((p.h + kWgRows - 1) / kWgRows) * p.n;
    const int kWgCi = 64 * bf16_wgrad_tiles_per_wave(ks);
    const int ci_groups = (p.cin + kWgCi - 1) / kWgCi, co_groups = ks == 3 ? (p.cout + 15) / 16 : (p.cout + kWgCo1 - 1) / kWgCo1;
    int blocks = (ks == 3 ? 512 : 256) / (ci_groups * co_groups);
    // (fewer, longer-running blocks at the coarse levels -- at least 8 tiles each -- halve the reduction but leave most CUs idle:
    // measured 11.7 vs 8.3 ms per step for the two kernels together, profiles/r03_r)
    blocks = blocks < 1 ? 1 : blocks;
    return blocks < tiles ? blocks : tiles;
}
inline int64_t bf16_wgrad_partial_floats(int cin, int cout, int ks, int blocks) {
    const int64_t ci_pad = (cin + 15) / 16 * 16, co_groups = ks == 3 ? (cout + 15) / 16 : (cout + kWgCo1 - 1) / kWgCo1;
    return blocks * co_groups * 9 * 16 * ci_pad;
}

template <int KS>
inline int launch_bf16_wgrad(Wgrad16Params p, float* dw, hipStream_t stream) {
    if ((p.cin & 3) || (p.cout & 3) || (p.ac0 & 7) || (p.gc0 & 3) || (p.a_blk & 7) || (p.g_blk & 3)) return ENDO_E_BADARG;
    if (KS == 1 && ((p.cout & 7) || (p.gc0 & 7))) return ENDO_E_BADARG;
    if (p.group_n > 0 && p.n > 2 * p.group_n) return ENDO_E_BADARG;          // two BatchNorm tables in LDS
    p.ci_pad = (p.cin + 15) / 16 * 16;
    const int blocks = bf16_wgrad_blocks(p, KS);
    if (p.partial_cap > 0 && bf16_wgrad_partial_floats(p.cin, p.cout, KS, blocks) > p.partial_cap) return ENDO_E_BADARG;   // never write past the workspace
    constexpr int T = bf16_wgrad_tiles_per_wave(KS), kWgCi = 64 * T;
    const int ci_groups = (p.cin + kWgCi - 1) / kWgCi, co_groups = KS == 3 ? (p.cout + 15) / 16 : (p.cout + kWgCo1 - 1) / kWgCo1;
    const size_t smem = bf16_wgrad_smem<KS>();
    ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bf16_wgrad_kernel<KS, T>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
    bf16_wgrad_kernel<KS, T><<<dim3(blocks, ci_groups, co_groups), 256, smem, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    const int64_t per_block = static_cast<int64_t>(co_groups) * 9 * 16 * p.ci_pad;
    bf16_wgrad_reduce_kernel<<<dim3(static_cast<int>((per_block + 255) / 256), (blocks + 63) / 64), 256, 0, stream>>>(p.partial, blocks, co_groups, p.ci_pad, p.cin, p.cout, KS, p.rot,
                                                                                            p.rot_n, dw, p.cin_w > 0 ? p.cin_w : p.cin, p.gscale);
    ENDO_LAUNCH_CHECK();
    return 0;
}

}  // inline namespace
}  // namespace endo
