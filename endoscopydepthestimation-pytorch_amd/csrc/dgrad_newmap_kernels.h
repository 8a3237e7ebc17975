// The "new-map" passes of a dense block's backward (the gradient into the 12 maps layer j-1 produced, from its NL = 1..3 consumers
// inside the block: reference models.py:33-62, DenseBlock.forward's concatenation seen from the back) as PERSISTENT blocks.
//
// dgrad_block_kernel<NL> (dgrad_block_kernels.h) runs these passes as one block per 32 x 6 pixel tile: 6 880 blocks at 16 x 256 x 320, each
// of which gathers its weight slices, derives its BN constants, waits for its NL * 12 gradient tiles, runs NL short steps with a barrier,
// an LDS reduction and 24 fp64 atomics each, and leaves.  Counters (profiles/r05_b_sq_counters.txt) and the diagnostic masks of
// tools/nl_bench say where that goes: the matrix pipe is busy 34 % of the time, a wave issues 5.6 vector and 4.5 scalar instructions per
// MFMA -- most of them block prologue and weight-gather addressing -- and the phases (tile load 64 us, MFMA 86, epilogue 60, operand loads
// and stores 40 of 251 us at NL = 3) ADD UP: with two or three waves per SIMD, each of them serial in itself, nothing overlaps.
//
// Here a block walks a contiguous run of tiles (grid = what is resident at once):
//   * once per block: the NL weight slices (gathered into LDS in fragment order), the BN constants of the NL layers (registers);
//   * per (tile, layer) CHUNK: the layer's 12 gradient maps of the haloed tile by 16-byte LDS-DMA into one of TWO stages -- the DMA of
//     chunk k+1 is issued right after the barrier that publishes chunk k, so a chunk's load has a whole chunk of MFMAs to land;
//   * the tile's x / old-gradient operands are requested one tile ahead, its results stored one chunk late (no load or store is waited for
//     in the chunk that issues it);
//   * the BN-backward sums stay in registers (fp32 per tile, fp64 across tiles) and leave by 24 NL atomics per BLOCK at the end.
// 16.1 KB per stage + 6.9 KB per layer of weights: 3 blocks per CU at NL = 2 / 3, 4 at NL = 1.
#pragma once

#include "dgrad_block_kernels.h"

namespace endo {

template <int NL>
struct NewMapGeom {
    using G = DgradBlockGeom<1, 2, 3, 1, 4>;                    // 32 x 6 tile, 40 x 8 haloed window, 16-byte DMA units
    static constexpr int kStage = 12 * G::kCS;                   // floats per chunk
    static constexpr int kW = 9 * 12 * 16;                       // floats per layer: [tap][c][16]
    static constexpr int kWPre = (kW + kConvThreads - 1) / kConvThreads;
    static constexpr int kConsts = NL * 16 * 4;                  // (scale, beta, mean, rstd) per layer and channel
    static constexpr size_t kBytes = sizeof(float) * (2 * kStage + NL * kW + kConsts);
    static constexpr int kBlocksPerCu = NL == 1 ? 4 : 3;
    static_assert(kBytes * kBlocksPerCu <= 160 * 1024, "LDS of the resident blocks");
    static_assert(sizeof(double) * 4 * NL * 16 * 2 <= sizeof(float) * kStage, "the final reduction aliases stage 0");
};

template <int NL>
__global__ void __launch_bounds__(kConvThreads, NewMapGeom<NL>::kBlocksPerCu) dgrad_newmap_kernel(const DgradBlockParams p, int tiles_y, int blocks_per_group, int tiles_per_block) {
    using NG = NewMapGeom<NL>;
    using G = typename NG::G;
    constexpr int R = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_g = smem;                               // [2][12][kCS]
    float* s_w = smem + 2 * NG::kStage;              // [NL][9][12][16]
    float* s_c = s_w + NL * NG::kW;                  // [NL][16][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int wx = (wave & 1) * 16;
    const int wy = (wave >> 1) * R;

    // ---- this block's tiles: all inside one group of the batch (BN statistics are per group) ----
    // The blocks of an XCD share one contiguous range of tiles and walk it interleaved (block j: tiles first + j, first + j + Q, ...): at any
    // time the XCD's blocks work on neighbouring tiles, whose haloed gradient windows then meet in the XCD's L2 (round 6: with a contiguous
    // run per block the window lines shared with the block's own next tile had left the L2 by the time it got there -- dgrad_wino3p_kernels.h)
    const int gn = p.group_n > 0 ? p.group_n : p.n;
    const int tps = p.tiles_x * tiles_y;
    const int Q = static_cast<int>(gridDim.x >> 3);
    const bool interleave = (gridDim.x & 7) == 0 && Q > 0 && blocks_per_group % Q == 0;
    int grp, first, t_step, t_last;          // first tile, stride, end of the range (tile indices inside the group)
    if (interleave) {
        const int lb0 = static_cast<int>(blockIdx.x & 7) * Q;
        grp = lb0 / blocks_per_group;
        const int b0 = lb0 - grp * blocks_per_group;
        first = b0 * tiles_per_block + static_cast<int>(blockIdx.x >> 3);
        t_last = min((b0 + Q) * tiles_per_block, tps * gn);
        t_step = Q;
    } else {
        const int lb = blockIdx.x;
        grp = lb / blocks_per_group;
        first = (lb - grp * blocks_per_group) * tiles_per_block;
        t_last = min(first + tiles_per_block, tps * gn);
        t_step = 1;
    }
    const int ntiles = first < t_last ? (t_last - first + t_step - 1) / t_step : 0;
    if (ntiles <= 0) return;
    const int64_t grp_off = p.group_n > 0 ? grp * p.gs : 0;

    struct Tile { int n, tx, ty, idx; };
    auto place = [&](Tile& t) {
        t.n = t.idx / tps;
        const int rem = t.idx - t.n * tps;
        t.ty = rem / p.tiles_x;
        t.tx = rem - t.ty * p.tiles_x;
    };
    auto advance = [&](Tile& t) { t.idx += t_step; place(t); };
    Tile cur;
    cur.idx = first;
    place(cur);

    // ---- weight slices, once: element (l, tap, c, j) <- W_l[c][w_ci_off + j][8 - tap] ----
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const float* wl = p.wgt[l] + static_cast<int64_t>(p.w_ci_off) * 9;
        const int wcin = p.w_cin[l];
#pragma unroll
        for (int k = 0; k < NG::kWPre; ++k) {
            const int e0 = k * kConvThreads + wave * 64;
            if (e0 < NG::kW) {
                const int e = e0 + lane;
                const int j = e & 15, rest = e >> 4;
                const int tap = rest / 12, cc = rest - tap * 12;
                const bool ok = e < NG::kW && j < p.count;
                const float* src = ok ? wl + cc * 9 * wcin + j * 9 + (8 - tap) : g_pad_consts + 4;
                if (e < NG::kW) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_w + l * NG::kW + e0), 4, 0, 0);
            }
        }
    }
    // ---- BN constants of the 16 channels of every layer: (scale, beta, mean, rstd), one 16-byte read per lane and chunk ----
    const bool co_ok = li < p.count;
    if (tid < NL * 16) {
        const int l = tid >> 4, j = tid & 15;
        const float* sv = p.saved[0];
        const float* ga = p.gamma[0];
        const float* be = p.beta[0];
        if (NL > 1 && l == 1) { sv = p.saved[1]; ga = p.gamma[1]; be = p.beta[1]; }
        if (NL > 2 && l == 2) { sv = p.saved[2]; ga = p.gamma[2]; be = p.beta[2]; }
        f32x4 c4{0.f, 0.f, 0.f, 0.f};
        if (j < p.count) {
            const float mean = sv[grp_off + 2 * j], rstd = sv[grp_off + 2 * j + 1];
            c4 = f32x4{ga[j] * rstd, be[j], mean, rstd};
        }
        *reinterpret_cast<f32x4*>(s_c + tid * 4) = c4;          // (published by the first chunk's barrier)
    }
    const float wfe = p.vg ? (co_ok ? p.vw[li] : 0.f) : 1.f;          // old gradient = wfe * (what the load returned)

    // ---- DMA units of this lane inside the 40 x 8 window: a DMA is descriptor (the sample's NL * 12 maps) + a per-lane byte offset that
    //      changes per tile + the map's scalar offset; out-of-image units read zeros through the descriptor's range check ----
    int u_ry[2], u_rx[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = lane + 64 * k;
        u_ry[k] = u / (G::kCols / 4);
        u_rx[k] = (u - u_ry[k] * (G::kCols / 4)) * 4;
    }
    const bool second_unit = lane + 64 < G::kUnits;
    const unsigned kOob = 0x80000000u;
    unsigned d_vo[2] = {kOob, kOob};
    const float* d_gn = p.g;
    auto dma_tile = [&](const Tile& t) {
        const int x0 = t.tx * G::kTileX, y0 = t.ty * G::kTileY;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int gy = y0 - 1 + u_ry[k], gx = x0 - G::kLeft + u_rx[k];
            const bool ok = gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
            d_vo[k] = ok ? 4u * static_cast<unsigned>(gy * p.g_w + gx) : kOob;
        }
        d_gn = p.g + grp_off + static_cast<int64_t>(t.n) * p.g_ns;
    };
    auto issue_chunk = [&](int l, int stage) {
        float* dst = s_g + stage * NG::kStage;
        const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d_gn), 0, NL * 12 * p.g_cs * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = wave + 4 * i;
            const unsigned so = 4u * static_cast<unsigned>((l * 12 + c) * p.g_cs);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lptr_t)(dst + c * G::kCS), 16, d_vo[0], so, 0, 0);
            if (second_unit) __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lptr_t)(dst + c * G::kCS + 256), 16, d_vo[1], so, 0, 0);
        }
    };

    // ---- epilogue operands of a tile: x and the old gradient are requested at the tile's first chunk (x is needed when its MFMAs are
    //      done, the old gradient when the last layer is); the results leave from the old gradient's registers one chunk late.
    //      Address = wave-uniform tile base (scalar) + a per-lane element offset that never changes ----
    f32x4 xc[R], dc[R], total[R];
    const int lane_x = wx + 4 * lk;
    unsigned lo_r[R];          // (channel li, row r of this wave, column lane_x) relative to the tile's first pixel
#pragma unroll
    for (int r = 0; r < R; ++r) lo_r[r] = static_cast<unsigned>(li * p.cs + (wy + r) * p.w + lane_x);
    const unsigned lo_ch = static_cast<unsigned>(li * p.cs);
    int64_t cur_off = 0, po_off = 0, cur_voff = 0;          // uniform: the tile's first pixel inside x / out, inside the virtual plane
    unsigned cur_rows = 0, po_rows = 0;                     // bit r: the lane's row r is inside the image (and its channel and columns exist)
    // A tile that lies inside the image (all but the bottom / right rim) takes the branch-free path: every lane loads and computes -- lanes
    // of channels 12..15 read the maps BEHIND the target's (a consumer's, they exist) and get zeros from the zero weight columns; their
    // sums and results are never looked at -- and only the stores are masked.  Rim tiles mask per row and lane.
    bool cur_full = false, po_full = false;                 // (uniform)
    auto begin_tile = [&](const Tile& t) {
        const int x0 = t.tx * G::kTileX, y0 = t.ty * G::kTileY;
        cur_full = y0 + G::kTileY <= p.h && x0 + G::kTileX <= p.w;
        cur_off = grp_off + static_cast<int64_t>(t.n) * p.ns + y0 * p.w + x0;
        cur_voff = grp_off + static_cast<int64_t>(t.n) * p.cs + y0 * p.w + x0;
        const float* xb = p.x + cur_off;
        const float* ob = p.vg ? p.vg + cur_voff : p.out + cur_off;          // (block-uniform) virtual: g * w_final[channel], one plane per sample
        if (cur_full) {
            cur_rows = co_ok ? 7u : 0u;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                xc[r] = *reinterpret_cast<const f32x4*>(xb + lo_r[r]);
                dc[r] = *reinterpret_cast<const f32x4*>(ob + (p.vg ? lo_r[r] - lo_ch : lo_r[r]));
            }
        } else {
            unsigned rows = 0;          // (uniform)
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (y0 + wy + r < p.h) rows |= 1u << r;
            cur_rows = (co_ok && x0 + lane_x + 3 < p.w) ? rows : 0u;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                xc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                dc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (cur_rows & (1u << r)) {
                    xc[r] = *reinterpret_cast<const f32x4*>(xb + lo_r[r]);
                    dc[r] = *reinterpret_cast<const f32x4*>(ob + (p.vg ? lo_r[r] - lo_ch : lo_r[r]));
                }
            }
        }
    };

    auto store_results = [&]() {
        float* ob = p.out + po_off;
        if (po_full) {
            if (co_ok) {
#pragma unroll
                for (int r = 0; r < R; ++r) *reinterpret_cast<f32x4*>(ob + lo_r[r]) = dc[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (po_rows & (1u << r)) *reinterpret_cast<f32x4*>(ob + lo_r[r]) = dc[r];
        }
    };

    // BN-backward sums: fp32 per lane over its 12 pixels of at most kFlushTiles tiles (<= 96 terms, fewer than one block of
    // dgrad_block_kernel adds up in fp32), then into fp64 registers: the run of a block grows with image size and batch, the fp32 part does not
    constexpr int kFlushTiles = 8;
    float fs1[NL], fs2[NL];
    double ds1[NL], ds2[NL];
    int tiles_since_flush = 0;
#pragma unroll
    for (int l = 0; l < NL; ++l) { fs1[l] = fs2[l] = 0.f; ds1[l] = ds2[l] = 0.0; }
    dma_tile(cur);
    issue_chunk(0, 0);
    Tile nxt = cur;
    advance(nxt);
    int stage = 0;

    for (int t = 0; t < ntiles; ++t) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this chunk's maps (and, the first time, the weights) have landed
            __syncthreads();
            if (l == 0) {
                if (t > 0) {          // the previous tile's results, one chunk late
                    store_results();
                }
                begin_tile(cur);
            }
            if (l + 1 < NL) {
                issue_chunk(l + 1, stage ^ 1);
            } else if (t + 1 < ntiles) {
                dma_tile(nxt);
                issue_chunk(0, stage ^ 1);
            }
            // ---- convT_l(G_l): K = 3 map quads x 9 taps, fragments of quad q+1 requested before the MFMAs of quad q ----
            f32x4 acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                const float* sg = s_g + stage * NG::kStage;
                const float* wb = s_w + l * NG::kW;
                float av[2][3][R + 2], bw[2][9];
                auto load_quad = [&](int quad, int set) {
                    const float* a_base = sg + (quad * 4 + lk) * G::kCS + wy * G::kCols + wx + li + G::kColOff;
                    const float* b_base = wb + (quad * 4 + lk) * 16 + li;
#pragma unroll
                    for (int r = 0; r < R + 2; ++r)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) av[set][dx][r] = a_base[r * G::kCols + dx];
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) bw[set][tap] = b_base[tap * 12 * 16];
                };
                load_quad(0, 0);
#pragma unroll
                for (int quad = 0; quad < 3; ++quad) {
                    const int set = quad & 1;
                    if (quad + 1 < 3) load_quad(quad + 1, set ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[set][dx][r + dy], bw[set][dy * 3 + dx], acc[r], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- layer l's ReLU mask + BN backward: dz = [z > 0] acc, sums of dz and dz (x - mean) (times rstd once per tile), the gradient
            //      scale_l dz added up over the layers ----
            {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(s_c + (l * 16 + li) * 4);
                const float scale = c4[0], beta = c4[1], mean = c4[2], rstd = c4[3];
                float s1 = 0.f, s2 = 0.f;
                auto row = [&](int r) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xcen = xc[r][e] - mean;
                        const float z = fmaf(xcen, scale, beta);
                        const float dz = z > 0.f ? acc[r][e] : 0.f;
                        s1 += dz;
                        s2 = fmaf(dz, xcen, s2);
                        total[r][e] = l == 0 ? scale * dz : fmaf(scale, dz, total[r][e]);
                    }
                };
                if (cur_full) {
#pragma unroll
                    for (int r = 0; r < R; ++r) row(r);
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if (l == 0) total[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (cur_rows & (1u << r)) row(r);
                    }
                }
                fs1[l] += s1;
                fs2[l] = fmaf(s2, rstd, fs2[l]);
            }
            if (l == NL - 1) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float old = dc[r][e] * wfe;
                        asm volatile("" : "+v"(old));          // no contraction into an fma: the product is rounded as a materialised g * w[c] is
                        dc[r][e] = old + total[r][e];
                    }
                po_off = cur_off;
                po_rows = cur_rows;
                po_full = cur_full;
                cur = nxt;
                advance(nxt);
                if (++tiles_since_flush == kFlushTiles) {
                    tiles_since_flush = 0;
#pragma unroll
                    for (int k = 0; k < NL; ++k) {
                        ds1[k] += static_cast<double>(fs1[k]); ds2[k] += static_cast<double>(fs2[k]);
                        fs1[k] = 0.f; fs2[k] = 0.f;
                    }
                }
            }
            stage ^= 1;
            // keep the stage a run-time value: with an even NL it is a compile-time constant per unrolled layer, and that build computed wrong
            // results on the GPU (tools/nl_bench, NL = 2; LDS written by DMA is invisible to the compiler) -- every other DMA pipeline of
            // this library indexes its buffers at run time as well
            asm volatile("" : "+s"(stage));
        }
    }
    store_results();

    // ---- BN-backward sums of the whole run: lanes -> waves -> block -> one fp64 atomic per (layer, channel, sum) ----
    __syncthreads();          // every wave is done with stage 0
    double* red = reinterpret_cast<double*>(smem);          // [4 waves][NL][16][2]
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        double a = ds1[l] + static_cast<double>(fs1[l]), b = ds2[l] + static_cast<double>(fs2[l]);
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
        if (lk == 0) {
            red[((wave * NL + l) * 16 + li) * 2] = a;
            red[((wave * NL + l) * 16 + li) * 2 + 1] = b;
        }
    }
    __syncthreads();
    if (tid < 32 * NL) {
        const int l = tid >> 5, j = (tid >> 1) & 15, which = tid & 1;
        if (j < p.count) {
            double t = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) t += red[((wv * NL + l) * 16 + j) * 2 + which];
            double* sc = p.scratch[0];          // (no dynamic index into the kernel argument)
            if (NL > 1 && l == 1) sc = p.scratch[1];
            if (NL > 2 && l == 2) sc = p.scratch[2];
            atomicAdd(sc + bn_slot_offset(p.slot_stride) + grp_off / 2 + 2 * j + which, t);
        }
    }
}

// the shapes the persistent form is written for: float4 rows (the 16-byte DMA and the float4 epilogue), one 16-channel group
// (the gradient maps of a sample and the x / out tile offsets go through 32-bit byte offsets: larger frames keep the per-tile pointer kernels)
inline bool dgrad_newmap_ok(const DgradBlockParams& p) {
    return p.count <= 16 && p.acc_from == 0 && 36ll * p.g_cs * 4 < (1ll << 31) && (16ll * p.cs + static_cast<int64_t>(p.h) * p.w) * 4 < (1ll << 32);
}

template <int NL>
inline int launch_dgrad_newmap(DgradBlockParams p, hipStream_t stream) {
    using NG = NewMapGeom<NL>;
    using G = typename NG::G;
    p.tiles_x = (p.w + G::kTileX - 1) / G::kTileX;
    const int tiles_y = (p.h + G::kTileY - 1) / G::kTileY;
    static bool configured_by_device[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& configured = configured_by_device[dev & 15];
    if (!configured && NG::kBytes > 48 * 1024) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_newmap_kernel<NL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(NG::kBytes)));
        configured = true;
    }
    const int groups = p.group_n > 0 ? p.n / p.group_n : 1;
    const int tiles_per_group = p.tiles_x * tiles_y * (p.group_n > 0 ? p.group_n : p.n);
    static int cus_by_device[16] = {};
    int& cus = cus_by_device[dev & 15];
    if (cus == 0) {
        hipDeviceProp_t prop;
        ENDO_CHECK(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount;
    }
    const int resident = cus * NG::kBlocksPerCu / groups;          // blocks per group that are on the chip at once
    const int tiles_per_block = (tiles_per_group + resident - 1) / resident;
    int blocks_per_group = (tiles_per_group + tiles_per_block - 1) / tiles_per_block;
    blocks_per_group = (blocks_per_group + 7) / 8 * 8;          // (empty blocks leave at once) a multiple of 8 for the XCD map
    dgrad_newmap_kernel<NL><<<dim3(groups * blocks_per_group), kConvThreads, NG::kBytes, stream>>>(p, tiles_y, blocks_per_group, tiles_per_block);
    ENDO_LAUNCH_CHECK();
    return 0;
}

}  // namespace endo
