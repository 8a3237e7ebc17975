// 3x3 weight gradient of the growth-12 dense layers, input channels split over WAVES (full-resolution levels).
//
//   dW[co][ci][ky][kx] = sum_p a[ci][p] * dY[co][p - (ky-1, kx-1)]          (a = relu(bn(x)), zero outside)
//
// Same GEMM as wgrad_taps_kernels.h (M = (co, tap) = 108 rows -> 7 MFMA row groups, N = input channels,
// K = pixels) but a block owns ALL input channels of a pass (its 16-channel groups dealt out to the 4 waves, at most
// NG each; when the count is not a multiple of 4 the waves holding one group more rotate with the block index so
// that the SIMDs of a CU, which run wave i of every resident block, stay evenly loaded) for its pixels:
//   * every activation value is used by exactly one lane (B[k = pixel][j = ci]), so x never touches LDS: each
//     lane loads its own 2 x 16 bytes per channel group straight into registers, one chunk ahead, and applies
//     BN+ReLU there;
//   * only the small operand -- 12 dY maps of a 3-row x 40-column window, 5.6 KiB -- is staged (LDS-DMA, two
//     buffers, one barrier per chunk) and shared by the 4 waves: 7 LDS reads feed 7*NG MFMAs;
//   * x and dY are read from HBM once per pass (the taps kernel re-reads the dY tile for every 16-channel slice);
//   * no cross-wave reduction: a wave owns its (row group, channel group) accumulators.  Blocks own contiguous
//     chunk ranges and write their partial sums to scratch (coalesced 256-byte rows); a small second kernel adds
//     the partials and accumulates into the flat gradient -- no same-address atomic storm, and the sum over
//     pixels is evaluated in a fixed order.
// k <-> pixel mapping inside a 32-pixel chunk row: k-step ks, lane group lk -> pixel 16*(ks>>2) + 4*lk + (ks&3),
// i.e. each lane's 8 pixels are two aligned float4s.
#pragma once

#include <type_traits>

#include "conv_dma_kernels.h"
#include "wgrad_kernels.h"

namespace endo {

constexpr int kNsSeg = 32;                         // pixels per chunk (one row segment)
constexpr int kNsCols = kNsSeg + 8;                // dY window: 4-pixel aligned halo on both sides
constexpr int kNsMap = 3 * kNsCols;                // floats per dY map window
constexpr int kNsBuf = 12 * kNsMap + 64;           // + zero rows read by the unused M rows of the last group
constexpr int kNsMG = 7;
constexpr int kNsUnits = 12 * kNsMap / 4;          // 360 float4

// EXP: diagnostic bit mask for tools/conv_bench (0 in the library): 1 = no x loads, 2 = no dY DMA, 4 = no barrier,
// 8 = fragment reads at consecutive (conflict-free, wrong) LDS addresses: what the 61 % bank conflicts of the real gathers cost
// BF: 1 = bf16 MFMA operands (fp32 accumulation, fp32 memory): the 8 k-steps of a lane's two float4s become one v_mfma_f32_16x16x32_bf16;
// 2 = fp32 operands as three bf16 terms each, six such MFMAs (fp32-accurate: common.h split_bf16x8)
template <int NG, int EXP = 0, int BF = 0>
__global__ void __launch_bounds__(kConvThreads) wgrad_nsplit_kernel(const WgradParams p, float* __restrict__ partial,
                                                                    int chunks_per_block) {
    __shared__ __attribute__((aligned(16))) float smem[3 * kNsBuf];      // dY windows of the chunk computed and the two in flight
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;
    // this pass's groups [pass_g0, pass_g0 + pass_n) dealt to the waves: rotated wave r owns q (+1 if r < rem) groups
    const int groups_total = (p.cin + 15) / 16;
    const int per_pass = (groups_total + gridDim.y - 1) / gridDim.y;
    const int pass_g0 = blockIdx.y * per_pass;
    const int pass_n = min(per_pass, groups_total - pass_g0);
    const int rw = (wave + blockIdx.x) & 3;
    const int gq = pass_n >> 2, grem = pass_n & 3;
    const int ngw = gq + (rw < grem ? 1 : 0);                 // groups of this wave (<= NG)
    const int group0 = pass_g0 + rw * gq + min(rw, grem);
    const int segs = (p.w + kNsSeg - 1) / kNsSeg;
    const int chunks_total = segs * p.h * p.n;
    const int c_begin = blockIdx.x * chunks_per_block;
    const int c_end = min(c_begin + chunks_per_block, chunks_total);

    float sc[NG], mn[NG], bt[NG];
    bool ch_ok[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) ch_ok[g] = g < ngw && 16 * (group0 + g) + li < p.cin;
    int cst_grp = -1;                          // sample group whose BN constants sit in sc / mn / bt
    auto load_consts = [&](int sg) {
        const float* saved = p.saved + sg * p.gs;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ch = 16 * (group0 + g) + li;
            sc[g] = 0.f; mn[g] = 0.f; bt[g] = 0.f;
            if (ch_ok[g]) {
                mn[g] = saved[2 * ch];
                sc[g] = p.gamma[ch] * saved[2 * ch + 1];
                bt[g] = p.beta[ch];
            }
        }
        cst_grp = sg;
    };

    // per-lane gather offsets into the dY window: row m = 16 g + li = co * 9 + ky * 3 + kx reads
    // dY[co][y + 1 - ky][x + 1 - kx]  ->  window row 2 - ky, window col x + 5 - kx
    int aoff[kNsMG];
#pragma unroll
    for (int g = 0; g < kNsMG; ++g) {
        const int m = 16 * g + li;
        const int co = m / 9, tap = m - co * 9;
        const int ky = tap / 3, kx = tap - ky * 3;
        aoff[g] = (m < 108) ? co * kNsMap + (2 - ky) * kNsCols + (5 - kx) + 4 * lk : 12 * kNsMap + 4 * lk;
    }
    for (int i = tid; i < 64; i += kConvThreads) {
        smem[12 * kNsMap + i] = 0.f;
        smem[kNsBuf + 12 * kNsMap + i] = 0.f;
        smem[2 * kNsBuf + 12 * kNsMap + i] = 0.f;
    }

    f32x4 acc[NG][kNsMG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int m = 0; m < kNsMG; ++m) acc[g][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* pad_zero = g_pad_consts + 4;
    // Loads run TWO chunks ahead of the MFMAs (one chunk of MFMAs, ~2.4 us, does not always cover an HBM round trip
    // under load): raw x of chunks c+1 and c+2 sit in xr[(c+1)&1], xr[c&1]; their dY windows in two of the three buffers.
    f32x4 xr[2][NG][2];
    unsigned xr_ok[2] = {0, 0};       // bit q: this lane's float4 q is inside the image

    // Everything about a chunk's loads that does not depend on the chunk is computed once: this thread's two units of
    // the dY window (map, window row, column) and this lane's channel rows.  The chunk position (sample, row, segment)
    // advances incrementally -- no divisions, no per-load branches in the loop (out-of-image units read a zero pad).
    int d_off[2], d_row[2], d_col[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = k * kConvThreads + tid;
        const int map = e / (kNsMap / 4);
        const int r = e - map * (kNsMap / 4);
        d_row[k] = r / (kNsCols / 4);
        d_col[k] = 4 * (r - d_row[k] * (kNsCols / 4));
        d_off[k] = map * p.dy_cs + d_row[k] * p.dy_w + d_col[k];
    }
    int64_t x_off[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) x_off[g] = static_cast<int64_t>(16 * (group0 + g) + li) * p.in_cs + 4 * lk;

    // position of the next chunk to issue
    int i_n = c_begin / (segs * p.h);
    int i_y = (c_begin - i_n * segs * p.h) / segs;
    int i_seg = c_begin - (i_n * p.h + i_y) * segs;
    // sample group of the chunk being computed (for the BN constants); follows the same walk one chunk behind
    int c_n = i_n, c_y = i_y, c_seg = i_seg;

    auto issue = [&](int buf, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const int x0 = i_seg * kNsSeg;
        const WgSample sm(p, i_n);
        float* s_dy = smem + buf * kNsBuf;
        const float* dy_base = p.dy + sm.dy_off(p) + static_cast<int64_t>(i_y - 1) * p.dy_w + x0 - 4;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e0 = k * kConvThreads + wave * 64;
            if (e0 < kNsUnits) {
                const bool ok = static_cast<unsigned>(i_y - 1 + d_row[k]) < static_cast<unsigned>(p.h) &&
                                static_cast<unsigned>(x0 - 4 + d_col[k]) < static_cast<unsigned>(p.w);
                const float* src = ok ? dy_base + d_off[k] : pad_zero;
                if (!(EXP & 2) && e0 + lane < kNsUnits) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_dy + 4 * e0), 16, 0, 0);
            }
        }
        const float* in_base = p.in + sm.in_off(p) + static_cast<int64_t>(i_y) * p.in_w + x0;
        unsigned okbits = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (x0 + 16 * q + 4 * lk < p.w) okbits |= 1u << q;
        xr_ok[slot] = okbits;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool ok = !(EXP & 1) && ch_ok[g] && (okbits & (1u << q));
                const float* src = ok ? in_base + x_off[g] + 16 * q : pad_zero;          // pad: 4 zeros, masked again after BN
                xr[slot][g][q] = *reinterpret_cast<const f32x4*>(src);
            }
        if (++i_seg == segs) {
            i_seg = 0;
            if (++i_y == p.h) { i_y = 0; ++i_n; }
        }
    };

    // every issue() is NG * 2 global loads plus this wave's share of the window DMA: waves 0 and 1 move two units each
    // (360 float4 over 256 threads), waves 2 and 3 one -- the count the vmcnt below leaves in flight
    using Slot0 = std::integral_constant<int, 0>;
    using Slot1 = std::integral_constant<int, 1>;
    if (c_begin < c_end) issue(0, Slot0{});
    if (c_begin + 1 < c_end) issue(1, Slot1{});
    int buf = 0;
    // one chunk; the register slot of its x values is a compile-time constant (the loop below alternates)
    auto do_chunk = [&](int chunk, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        if (chunk + 1 < c_end) {          // chunk + 1's loads may stay in flight
            if ((EXP & 2) || wave >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG * 2 + ((EXP & 2) ? 0 : 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NG * 2 + 2) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!(EXP & 4)) __syncthreads();
        // BN + ReLU of this chunk's activations (registers), then start the loads of chunk + 2 into the slot just freed
        {
            const int sg = WgSample(p, c_n).grp;
            if (sg != cst_grp) load_consts(sg);
            if (++c_seg == segs) {
                c_seg = 0;
                if (++c_y == p.h) { c_y = 0; ++c_n; }
            }
        }
        f32x4 bv[NG][2];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool ok = ch_ok[g] && (xr_ok[slot] & (1u << q));
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[g][q][e] = ok ? __builtin_fmaxf(fmaf(xr[slot][g][q][e] - mn[g], sc[g], bt[g]), 0.f) : 0.f;
            }
        if (chunk + 2 < c_end) issue(buf == 0 ? 2 : buf - 1, slot_c);          // the buffer computed last iteration

        const float* s_dy = smem + buf * kNsBuf;
        if constexpr (BF == 2) {
            // fp32 products on the bf16 matrix cores (common.h, split_bf16x8): both operands split into three bf16 terms, six MFMAs per
            // (row group, channel group) in place of the eight fp32 ones -- 96 instead of 256 matrix-pipe cycles, off the vector lanes
            Bf16x8Split bq[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g)
                bq[g] = split_bf16x8(bv[g][0][0], bv[g][0][1], bv[g][0][2], bv[g][0][3], bv[g][1][0], bv[g][1][1], bv[g][1][2], bv[g][1][3]);
#pragma unroll
            for (int m = 0; m < kNsMG; ++m) {
                const float* ap = s_dy + aoff[m];
                const Bf16x8Split aq = split_bf16x8(ap[0], ap[1], ap[2], ap[3], ap[16], ap[17], ap[18], ap[19]);
                // smallest products first; consecutive MFMAs go to different accumulators
#pragma unroll
                for (int g = 0; g < NG; ++g) if (g < ngw) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq.lo, bq[g].hi, acc[g][m], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) if (g < ngw) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq.hi, bq[g].lo, acc[g][m], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) if (g < ngw) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq.mid, bq[g].mid, acc[g][m], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) if (g < ngw) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq.mid, bq[g].hi, acc[g][m], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) if (g < ngw) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq.hi, bq[g].mid, acc[g][m], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) if (g < ngw) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq.hi, bq[g].hi, acc[g][m], 0, 0, 0);
            }
        } else if constexpr (BF != 0) {
            // the lane's 8 pixels of the chunk (two float4s) are the k = 8 lk + i of ONE v_mfma_f32_16x16x32_bf16 per (row group, channel group)
            bf16x8_t bq[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g)
                bq[g] = pack_bf16x8(bv[g][0][0], bv[g][0][1], bv[g][0][2], bv[g][0][3], bv[g][1][0], bv[g][1][1], bv[g][1][2], bv[g][1][3]);
#pragma unroll
            for (int m = 0; m < kNsMG; ++m) {
                const float* ap = s_dy + aoff[m];
                const bf16x8_t aq = pack_bf16x8(ap[0], ap[1], ap[2], ap[3], ap[16], ap[17], ap[18], ap[19]);
#pragma unroll
                for (int rep = 0; rep < (BF == 3 ? 6 : 1); ++rep)          // BF = 3 (tools/x3_bench only): six MFMAs per pair on the ROUNDED operands -- what the matrix work of the x3 form costs without its splits
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    if (g < ngw) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq, bq[g], acc[g][m], 0, 0, 0);
            }
        } else
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a[kNsMG];
#pragma unroll
                for (int m = 0; m < kNsMG; ++m) a[m] = s_dy[((EXP & 8) ? lane + 64 * m : aoff[m]) + 16 * q + e];          // EXP 8: conflict-free (wrong) addresses
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    if (g < ngw) {          // wave-uniform
#pragma unroll
                        for (int m = 0; m < kNsMG; ++m)
                            acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bv[g][q][e], acc[g][m], 0, 0, 0);
                    }
            }
        buf = buf == 2 ? 0 : buf + 1;
    };
    for (int chunk = c_begin; chunk < c_end; chunk += 2) {
        do_chunk(chunk, Slot0{});
        if (chunk + 1 < c_end) do_chunk(chunk + 1, Slot1{});
    }

    // partial[(group * 7 + m) * blocks + block][r][lane] = D[row 16 m + 4 lk + r][ci = 16 group + li]: row-block major, so that the
    // reduce kernel streams one contiguous run of `blocks` 1 KB rows per (group, m) instead of gathering them 86 KB apart
    const int64_t nblocks = gridDim.x;
#pragma unroll
    for (int g = 0; g < NG; ++g)
        if (g < ngw) {
#pragma unroll
            for (int m = 0; m < kNsMG; ++m) {
                float* out = partial + ((static_cast<int64_t>(group0 + g) * kNsMG + m) * nblocks + blockIdx.x) * 256;
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r * 64 + lane] = acc[g][m][r];
            }
        }
}

// grid (groups_total * 7, slices): the partial rows of row block (group, m) are one contiguous run of `blocks` x 256 floats; a
// block of the grid adds its slice of them -- float4 per lane, four rows in flight per thread block pass, a fixed order per
// (slice, element) -- and adds the slice's sum to the flat gradient (one atomic per element and slice).
// ORDER: which (co, tap) a GEMM row m stands for -- 0: m = 9 co + tap (wgrad_nsplit_kernel), 1: m = 12 tap + co (wgrad_x3_kernel)
template <int ORDER = 0>
__global__ void __launch_bounds__(256) wgrad_nsplit_reduce_kernel(const float* __restrict__ partial, int blocks, int groups_total, int cin,
                                                                  float* __restrict__ dw) {
    __shared__ f32x4 s_part[4][64];
    const int gm = blockIdx.x;
    const int group = gm / kNsMG, m7 = gm - group * kNsMG;
    const int per = (blocks + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(blocks, b0 + per);
    if (b0 >= b1) return;
    const int q = threadIdx.x & 63, sub = threadIdx.x >> 6;          // float4 q of a row; rows b0 + sub, b0 + sub + 4, ...
    const f32x4* src = reinterpret_cast<const f32x4*>(partial + (static_cast<int64_t>(gm) * blocks) * 256) + q;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int b = b0 + sub;
    for (; b + 4 < b1; b += 8) {
        s0 += src[static_cast<int64_t>(b) * 64];
        s1 += src[static_cast<int64_t>(b + 4) * 64];
    }
    if (b < b1) s0 += src[static_cast<int64_t>(b) * 64];
    s_part[sub][q] = s0 + s1;
    __syncthreads();
    // element e = 4 q + k of the row: r = e >> 6, lane = e & 63  ->  thread t sums the four partial rows of element t
    const int e = threadIdx.x;
    const float* sp = reinterpret_cast<const float*>(s_part);
    const float total = (sp[e] + sp[256 + e]) + (sp[512 + e] + sp[768 + e]);
    const int lane = e & 63, r = e >> 6;
    const int m = 16 * m7 + 4 * (lane >> 4) + r;
    const int ci = 16 * group + (lane & 15);
    if (m < 108 && ci < cin) {
        const int co = ORDER == 0 ? m / 9 : m % 12, tap = ORDER == 0 ? m - co * 9 : m / 12;
        atomicAdd(dw + (static_cast<int64_t>(co) * cin + ci) * 9 + tap, total);
    }
}

constexpr int kNsMinChunks = 2048;           // launches with fewer row chunks go to the tap-folded kernel
#ifndef ENDO_NS_BLOCKS3
#define ENDO_NS_BLOCKS3 512                  // blocks of an NG = 3 launch (in-job A/B: 512 = 2 per CU, 768 = 3 per CU)
#endif
constexpr int kNsMaxBlocks = 1024;
constexpr int64_t kNsScratchFloats = static_cast<int64_t>(kNsMaxBlocks) * 12 * kNsMG * 256;   // blocks * groups <= 1024 * 12

inline bool wgrad_nsplit_ok(const WgradParams& p) {
    const bool aligned = (p.w % 4 == 0) && (p.dy_w % 4 == 0) && (p.dy_cs % 4 == 0) && (p.dy_ns % 4 == 0) && (p.in_w % 4 == 0) &&
                         (p.in_cs % 4 == 0) && (p.in_ns % 4 == 0) && (reinterpret_cast<uintptr_t>(p.dy) % 16 == 0) &&
                         (reinterpret_cast<uintptr_t>(p.in) % 16 == 0);
    const long chunks = static_cast<long>((p.w + kNsSeg - 1) / kNsSeg) * p.h * p.n;
    return aligned && p.cout == 12 && chunks >= kNsMinChunks;
}

template <int NG, int EXP = 0, int BF = 0>
inline int launch_wgrad_nsplit_ng(const WgradParams& p, float* scratch, int passes, hipStream_t stream) {
    const int chunks_total = ((p.w + kNsSeg - 1) / kNsSeg) * p.h * p.n;
    int blocks = (NG == 3 ? ENDO_NS_BLOCKS3 : NG == 2 ? 768 : 1024) / passes;   // resident blocks per CU by register count: 2 / 3 / 4
    const int per = (chunks_total + blocks - 1) / blocks;
    blocks = (chunks_total + per - 1) / per;
    const int groups_total = (p.cin + 15) / 16;
    wgrad_nsplit_kernel<NG, EXP, BF><<<dim3(blocks, passes), kConvThreads, 0, stream>>>(p, scratch, per);
    ENDO_LAUNCH_CHECK();
    wgrad_nsplit_reduce_kernel<0><<<dim3(groups_total * kNsMG, 16), 256, 0, stream>>>(scratch, blocks, groups_total, p.cin, p.dw);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// scratch: kNsScratchFloats floats.  mfma_mode: 0 = fp32 matrix instructions, 1 = operands ROUNDED to bf16 (the mixed-precision mode of
// DESIGN.md 4.10), 2 = fp32 operands split into three bf16 terms, six bf16 MFMAs (common.h: fp32-accurate products on the bf16 cores)
inline int launch_wgrad_nsplit(const WgradParams& p, float* scratch, hipStream_t stream, int mfma_mode = 0) {
    const int groups = (p.cin + 15) / 16;
    const int passes = (groups + 11) / 12;                       // at most 3 groups per wave
    const int per_pass = (groups + passes - 1) / passes;
    const int ng = (per_pass + 3) / 4;
    if (mfma_mode == 2) {
        if (ng <= 1) return launch_wgrad_nsplit_ng<1, 0, 2>(p, scratch, passes, stream);
        if (ng == 2) return launch_wgrad_nsplit_ng<2, 0, 2>(p, scratch, passes, stream);
        return launch_wgrad_nsplit_ng<3, 0, 2>(p, scratch, passes, stream);
    }
    if (mfma_mode == 3) {          // diagnostic (tools/x3_bench)
        if (ng <= 1) return launch_wgrad_nsplit_ng<1, 0, 3>(p, scratch, passes, stream);
        if (ng == 2) return launch_wgrad_nsplit_ng<2, 0, 3>(p, scratch, passes, stream);
        return launch_wgrad_nsplit_ng<3, 0, 3>(p, scratch, passes, stream);
    }
    if (mfma_mode == 1) {
        if (ng <= 1) return launch_wgrad_nsplit_ng<1, 0, 1>(p, scratch, passes, stream);
        if (ng == 2) return launch_wgrad_nsplit_ng<2, 0, 1>(p, scratch, passes, stream);
        return launch_wgrad_nsplit_ng<3, 0, 1>(p, scratch, passes, stream);
    }
    if (ng <= 1) return launch_wgrad_nsplit_ng<1>(p, scratch, passes, stream);
    if (ng == 2) return launch_wgrad_nsplit_ng<2>(p, scratch, passes, stream);
    return launch_wgrad_nsplit_ng<3>(p, scratch, passes, stream);
}

}  // namespace endo
