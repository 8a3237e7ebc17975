// Forward of a transition-down layer (reference models.py:56-67: BN -> ReLU -> conv1x1 -> [dropout] -> maxpool2) as PERSISTENT blocks
// (round 6).  Same function as conv_dma_kernel<1, 8, 3, IN_BNRELU, EPI_FWD_POOL>:
//     y[o][p] = bias[o] + sum over c of W[o][c] relu(bn(x[c][p]));   out[o][pp] = max over the 2x2 window, code = its argmax (2 dy + dx);
//     per-channel sum / sum^2 of the stored values for the BatchNorm layers that read them.
// The per-tile kernel runs 10 240 blocks at level 0 (320 tiles x 2 halves of the output channels x 16 samples): every block streams the
// tile's 96 input planes (twice per tile), gathers a 384-float weight slice per K-chunk by dword LDS-DMA (6 of a chunk's 14 DMA
// instructions, each as expensive for the CU's address path as a 16-byte one) and runs 388 us for 154 us of matrix work (660 MB: 130 us
// of HBM).  Here a block of 8 waves stays on its CU and walks 32 x 8 pixel tiles of one group of the batch:
//   * the C x C weights are LDS-resident (transposed: [c][o]) for the block's lifetime (C = 96 / 144: levels 0 / 1), row stride == 16 (mod 32) dwords;
//   * ALL output channels per tile: the tile's input planes are read once, in K-chunks of 16 channels (16 KB, one 16-byte DMA
//     instruction per channel, 2 per wave) through three LDS stages -- the chunk after next is in flight while one is multiplied; the
//     chunks run on across tile boundaries;
//   * a wave owns two rows x 16 pixels (2 MFMA row tiles: the rows of its 2x2 pooling windows) x all C / 16 column tiles;
//   * the LDS image of a 16-byte DMA is lane-linear and a channel's 8 x 32 pixels are exactly 256 dwords, so the two k-lanes of a 32-lane
//     fragment pass (channels 4 ks + {0, 1} / {2, 3}) would meet on the same banks: the SOURCE is swizzled instead of padding -- odd
//     channels are stored with 16-pixel halves exchanged (unit u at u ^ 4);
//   * BN + ReLU on the fragment read with the per-channel constants of conv_kernels.h's bn_input_constants (which also records
//     (mean, rstd) for the backward pass and updates the running statistics: the first block of each group / of the launch);
//   * the statistics of the stored values: per lane over the block's run, reduced once (waves in a fixed order, one fp64 atomic per
//     (channel, sum) and BLOCK).
#pragma once

#include "conv_dma_kernels.h"

namespace endo {

// WAVES = 8: one block per CU, a wave owns two rows x 16 pixels (2 MFMA row tiles), three LDS stages.  WAVES = 4 (C = 96 only: 77 KB of LDS): TWO
// blocks per CU, a wave owns two rows x 32 pixels (4 row tiles), two stages -- the blocks drift apart and one's epilogue / DMA waits
// run under the other's MFMAs (with one block per CU the phases of a tile add up: tools/td_bench's diagnostic builds)
template <int C, int WAVES = 8>
struct TdFwdGeom {
    static_assert(C % 16 == 0, "whole MFMA column tiles and K-chunks");
    static_assert(WAVES == 8 || WAVES == 4, "waves per block");
    static constexpr int kThreads = 64 * WAVES;
    static constexpr int kMT = 16 / WAVES;                              // MFMA row tiles per wave
    static constexpr int kCPW = 16 / WAVES;                             // channels of a K-chunk a wave DMAs
    static constexpr int kTileX = 32, kTileY = 8;
    static constexpr int kKC = 16;                                      // input channels per K-chunk
    static constexpr int kStages = WAVES == 8 ? 3 : 2;
    static constexpr int kChunkFloats = kKC * 256;
    static constexpr int kWS = (C % 32 == 16) ? C : C + 16;            // weight row stride == 16 (mod 32) dwords
    static constexpr int kNT = C / 16;                                  // column tiles
    static constexpr int kChunks = C / kKC;                             // K-chunks per tile
    static constexpr int kFloats = C * kWS + kStages * kChunkFloats + 4 * C + WAVES * C * 2;
    static constexpr size_t kBytes = sizeof(float) * kFloats;
    static_assert(kBytes <= (WAVES == 8 ? 160 : 80) * 1024, "blocks per CU");
};

// p0: the ConvParams of the per-tile launch (td_fwd).  tiles_xy = tiles per sample, gn = samples per group, bpg = blocks per group.
// EXP: diagnostic bit mask for tools/td_bench (0 in the library; timing only): 1 = no input DMA, 2 = no epilogue (stores, statistics), 4 = no MFMAs,
// 8 = no BN + ReLU on the fragment read
template <int C, int WAVES = 8, int EXP = 0>
__global__ void __launch_bounds__(64 * WAVES, 2) td_fwd_kernel(const ConvParams p0, int tiles_xy, int gn, int bpg) {
    using G = TdFwdGeom<C, WAVES>;
    const int groups = gridDim.x / bpg;
    const int grp = blockIdx.x / bpg;
    const int r0 = blockIdx.x - grp * bpg;
    const ConvParams p = group_view(p0, grp);
    const int t_total = tiles_xy * gn;
    // the blocks of an XCD (bpg / 8 per group) share one contiguous tile range and walk it interleaved (dgrad_wino3p_kernels.h): a tile's
    // pooled rows are 64 bytes and its code rows 16 bytes of 128-byte lines whose rest belongs to the tiles beside it
    int t_begin, t_end, t_step;
    if ((bpg & 7) == 0) {
        const int q8 = bpg >> 3, xcd = r0 & 7, idx = r0 >> 3;
        t_begin = static_cast<int>(static_cast<int64_t>(xcd * q8) * t_total / bpg) + idx;
        t_end = static_cast<int>(static_cast<int64_t>((xcd + 1) * q8) * t_total / bpg);
        t_step = q8;
    } else {
        t_begin = static_cast<int>(static_cast<int64_t>(r0) * t_total / bpg);
        t_end = static_cast<int>(static_cast<int64_t>(r0 + 1) * t_total / bpg);
        t_step = 1;
    }

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_w = smem;                                   // [c][kWS]: W[o][c] TRANSPOSED (a B fragment's lanes run over o: contiguous; its k-lanes over c: rows)
    float* s_x = s_w + C * G::kWS;                       // [stage][channel 16][256] (odd channels: 16-pixel halves exchanged)
    float* s_bn = s_x + G::kStages * G::kChunkFloats;    // [C][scale, mean, beta, -]: one 16-byte read per k-step
    float* s_sum = s_bn + 4 * C;                         // [wave][C][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;

    // ---- once per block: weights, BN constants (and their side effects), zeroed sums ----
    for (int i = tid; i < C * C / 4; i += G::kThreads) {
        const int o = i / (C / 4), c4 = (i - o * (C / 4)) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.wgt + static_cast<int64_t>(o) * p.w_cin + c4);
#pragma unroll
        for (int k = 0; k < 4; ++k) s_w[(c4 + k) * G::kWS + o] = v[k];
    }
    {
        const bool first_of_group = r0 == 0;
        for (int c = tid; c < C; c += G::kThreads) {
            float scale, mean, beta;
            bn_input_constants(p, p0, grp, groups, first_of_group, c, scale, mean, beta);
            *reinterpret_cast<f32x4*>(s_bn + 4 * c) = f32x4{scale, mean, beta, 0.f};
        }
    }
    for (int i = tid; i < WAVES * C * 2; i += G::kThreads) s_sum[i] = 0.f;
    if (t_begin >= t_end) return;          // (block-uniform)

    auto tile_origin = [&](int t, int& n, int& x0, int& y0) {
        n = t / tiles_xy;
        const int tile = t - n * tiles_xy;
        const int ty = tile / p.tiles_x;
        x0 = (tile - ty * p.tiles_x) * G::kTileX;
        y0 = ty * G::kTileY;
    };
    // ---- chunk q of the run (tile t_begin + q / kChunks, channels 16 (q % kChunks) ..) -> stage q % 3: wave w DMAs channels 2 w, 2 w + 1.
    //      lane = LDS unit (row lane >> 3, 4 pixels at 4 (lane & 7)); an odd channel takes its source from unit lane ^ 4 ----
    const unsigned src_even = 4u * static_cast<unsigned>((lane >> 3) * p.in_w + 4 * (lane & 7));
    const unsigned src_odd = 4u * static_cast<unsigned>((lane >> 3) * p.in_w + 4 * ((lane & 7) ^ 4));
    const int nq = ((t_end - t_begin + t_step - 1) / t_step) * G::kChunks;
    auto issue_chunk = [&](int q) {
        if constexpr ((EXP & 1) != 0) return;
        const int t = t_begin + (q / G::kChunks) * t_step, ch0 = (q % G::kChunks) * G::kKC;
        int n, x0, y0;
        tile_origin(t, n, x0, y0);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in + static_cast<int64_t>(n) * p.in_ns), 0, 0x7ffffffc, 0x00020000);
        float* dst = s_x + (q % G::kStages) * G::kChunkFloats + (G::kCPW * wave) * 256;
        unsigned so = 4u * static_cast<unsigned>((ch0 + G::kCPW * wave) * p.in_cs + y0 * p.in_w + x0);
#pragma unroll
        for (int i = 0; i < G::kCPW; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr_t)(dst + i * 256), 16, (i & 1) ? src_odd : src_even, so, 0, 0);
            so += 4u * static_cast<unsigned>(p.in_cs);
        }
    };

    // ---- fragment addressing: row tile mt of the wave = (row 2 rp + (mt & 1), 16-pixel column half chh); A for channel 4 ks + lk ----
    const int rp = WAVES == 8 ? wave >> 1 : wave;
    int a_off[G::kMT];
#pragma unroll
    for (int mt = 0; mt < G::kMT; ++mt) {
        const int chh = WAVES == 8 ? (wave & 1) : (mt >> 1);
        const int unit = (2 * rp + (mt & 1)) * 8 + 4 * chh + (li >> 2);
        a_off[mt] = lk * 256 + ((unit ^ (4 * (lk & 1))) * 4) + (li & 3);          // + ks * 1024
    }

    // statistics of the stored values: per lane over the block's whole run (fp32: 2 values per tile and column tile), reduced ONCE at the end --
    // per tile the four shuffles per column tile, the barrier and the atomics were 2 of a tile's 17 us (tools/td_bench: 347 -> 299 us without the epilogue's rest)
    float st1[G::kNT], st2[G::kNT];
#pragma unroll
    for (int nt = 0; nt < G::kNT; ++nt) { st1[nt] = 0.f; st2[nt] = 0.f; }
    issue_chunk(0);
    if (G::kStages == 3 && nq > 1) issue_chunk(1);
    int q = 0;
    for (int t = t_begin; t < t_end; t += t_step) {
        int n, x0, y0;
        tile_origin(t, n, x0, y0);
        f32x4 acc[G::kMT][G::kNT];
#pragma unroll
        for (int tt = 0; tt < G::kMT; ++tt)
#pragma unroll
            for (int nt = 0; nt < G::kNT; ++nt) acc[tt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kc = 0; kc < G::kChunks; ++kc, ++q) {
            // chunk q has landed (three stages: everything but the newest chunk's DMAs of this wave); the barrier publishes it and retires the
            // reads of chunk q - 1, whose stage the next chunk to be requested then takes
            if (G::kStages == 3 && q + 1 < nq) __builtin_amdgcn_s_waitcnt(0x0070 | G::kCPW); else __builtin_amdgcn_s_waitcnt(0x0070);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (q + G::kStages - 1 < nq) issue_chunk(q + G::kStages - 1);
            const float* sx = s_x + (q % G::kStages) * G::kChunkFloats;
            const int c0 = kc * G::kKC;
#pragma unroll
            for (int ks = 0; ks < G::kKC / 4; ++ks) {
                const int c = c0 + 4 * ks + lk;
                const f32x4 bn = *reinterpret_cast<const f32x4*>(s_bn + 4 * c);
                const float sc = bn[0], mn = bn[1], bt = bn[2];
                float a[G::kMT];
#pragma unroll
                for (int tt = 0; tt < G::kMT; ++tt) a[tt] = (EXP & 8) ? sx[a_off[tt] + ks * 1024] : __builtin_fmaxf(fmaf(sx[a_off[tt] + ks * 1024] - mn, sc, bt), 0.f);
#pragma unroll
                for (int nt = 0; nt < G::kNT; ++nt) {
                    const float b = s_w[c * G::kWS + nt * 16 + li];
#pragma unroll
                    for (int tt = 0; tt < G::kMT; ++tt) {
                        if constexpr ((EXP & 4) != 0) acc[tt][nt][0] += a[tt] * b;
                        else acc[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt], b, acc[tt][nt], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr ((EXP & 2) != 0) {
            float keep = 0.f;
#pragma unroll
            for (int nt = 0; nt < G::kNT; ++nt) keep += acc[0][nt][0] + acc[G::kMT - 1][nt][1];
            if (keep == 123.456f) p.out[0] = keep;
            continue;
        }
        // ---- epilogue: bias, 2x2 max-pool + argmax (per column half: the lane's 4 pixels of 2 rows = 2 windows), stores, statistics ----
        const int yp = (y0 >> 1) + rp;
        float* out_n = p.out + static_cast<int64_t>(n) * p.out_ns;
        uint8_t* idx_n = p.out_idx + static_cast<int64_t>(n) * p.idx_ns;
#pragma unroll
        for (int hb = 0; hb < G::kMT / 2; ++hb) {
            const int chh = WAVES == 8 ? (wave & 1) : hb;
            const int xp = (x0 >> 1) + 8 * chh + 2 * lk;
#pragma unroll
            for (int nt = 0; nt < G::kNT; ++nt) {
                const int co = nt * 16 + li;
                const float bias = p.bias ? p.bias[co] : 0.f;
                float best2[2];
                unsigned code2[2];
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    const int e = 2 * w2;
                    float best = acc[2 * hb][nt][e] + bias;
                    unsigned code = 0;
                    float v = acc[2 * hb][nt][e + 1] + bias;
                    if (v > best) { best = v; code = 1; }
                    v = acc[2 * hb + 1][nt][e] + bias;
                    if (v > best) { best = v; code = 2; }
                    v = acc[2 * hb + 1][nt][e + 1] + bias;
                    if (v > best) { best = v; code = 3; }
                    best2[w2] = best; code2[w2] = code;
                    st1[nt] += best; st2[nt] += best * best;
                }
                const int64_t o = static_cast<int64_t>(co) * p.out_cs + yp * p.out_w + xp;
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<f32x2*>(out_n + o) = f32x2{best2[0], best2[1]};
                *reinterpret_cast<uint16_t*>(idx_n + o) = static_cast<uint16_t>(code2[0] | (code2[1] << 8));
            }
        }
    }
    // ---- once per block: the statistics (rows of the wave, waves in a fixed order, one fp64 atomic per (channel, sum)) ----
    if (p.out_sums) {
#pragma unroll
        for (int nt = 0; nt < G::kNT; ++nt) {
            float s1 = st1[nt], s2 = st2[nt];
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lk == 0) { s_sum[(wave * C + nt * 16 + li) * 2] = s1; s_sum[(wave * C + nt * 16 + li) * 2 + 1] = s2; }
        }
        __syncthreads();
        if (tid < 2 * C) {
            double v = 0.0;
#pragma unroll
            for (int wv = 0; wv < WAVES; ++wv) v += static_cast<double>(s_sum[wv * C * 2 + tid]);
            atomicAdd(p.out_sums + tid, v);
        }
    }
}

inline bool td_fwd_ok(const ConvParams& p) {
    return (p.cin == 96 || p.cin == 144) && p.cout == p.cin && p.w_cin == p.cin && (p.w % 32) == 0 && (p.h % 8) == 0 && p.in_w == p.w &&
           (p.in_cs % 4) == 0 && (p.in_ns % 4) == 0 && p.out_w == p.w / 2 && (p.out_cs % 2) == 0 && (p.out_ns % 2) == 0 && (p.idx_ns % 2) == 0 &&
           p.out_idx != nullptr && (reinterpret_cast<uintptr_t>(p.in) % 16) == 0 && (reinterpret_cast<uintptr_t>(p.wgt) % 16) == 0 &&
           (reinterpret_cast<uintptr_t>(p.out) % 8) == 0 && (reinterpret_cast<uintptr_t>(p.out_idx) % 2) == 0 && p.ksplit == 0 &&
           static_cast<int64_t>(p.cin) * p.in_cs * 4 < (1ll << 31);
}

template <int C, int WAVES = 8, int EXP = 0>
inline int launch_td_fwd_t(ConvParams p, int blocks, hipStream_t stream) {
    using G = TdFwdGeom<C, WAVES>;
    p.tiles_x = p.w / G::kTileX;
    const int tiles_xy = p.tiles_x * (p.h / G::kTileY);
    const int groups = p.group_n > 0 ? p.n / p.group_n : 1;
    const int gn = p.group_n > 0 ? p.group_n : p.n;
    int bpg = blocks * (WAVES == 8 ? 1 : 2) / groups;          // one or two blocks per CU
    if (bpg >= 8) bpg &= ~7;
    if (bpg > tiles_xy * gn) bpg = tiles_xy * gn;
    if (bpg < 1) bpg = 1;
    ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(td_fwd_kernel<C, WAVES, EXP>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(G::kBytes)));
    td_fwd_kernel<C, WAVES, EXP><<<dim3(bpg * groups), G::kThreads, G::kBytes, stream>>>(p, tiles_xy, gn, bpg);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// blocks = compute units of the device
inline int launch_td_fwd(const ConvParams& p, int blocks, hipStream_t stream) {
    if (!td_fwd_ok(p)) return ENDO_E_UNSUPPORTED;
    return p.cin == 96 ? launch_td_fwd_t<96, 4>(p, blocks, stream) : launch_td_fwd_t<144, 8>(p, blocks, stream);
}

}  // namespace endo
