// Data gradient of a transition-down layer (reference models.py:56-67: BN -> ReLU -> conv1x1 -> [dropout] -> maxpool2) as PERSISTENT
// blocks (round 6).  What it computes is conv_dma_kernel<1, 16, 2, IN_UNPOOL, EPI_DGRAD_BN>'s function:
//     dA[c][p]  = sum over o of  W[o][c] * U[o][p],   U = the pooled gradient routed back through the stored 2x2 argmax
//     dz        = dA * [z > 0],  z = BN(x)[c][p];   S1[c] += dz,  S2[c] += dz * xhat;   out[c][p] (+)= gamma rstd dz
// The per-tile kernel runs 15 360 blocks at level 0 (320 tiles x 3 slices of 32 output channels x 16 samples), each of which stages
// the tile's pooled gradient and argmax codes again, by DWORD LDS-DMA (168 instructions per block: 2.6 M per launch, ~45 cycles of a
// CU's address path each -- 190 us of the launch's 555), gathers its weight slice per K-chunk, and reads x / reads-modifies-writes the
// gradient in a synchronous epilogue.  Here a block of 8 waves stays on its CU and walks 32 x 8 pixel tiles of one group of the batch:
//   * the C x C weights are LDS-resident for the block's lifetime (C = 96 / 144: levels 0 / 1; row stride == 16 (mod 32) dwords);
//   * a tile's pooled gradient (C x 16 x 4) and codes arrive by 16-BYTE LDS-DMA -- C / 4 + C / 16 instructions per tile.  The LDS
//     image of a DMA is lane-linear, so the padding that kept the fragment reads conflict-free is replaced by a swizzle of the SOURCE:
//     unit q of channel o holds the pooled pixels of unit q ^ 2 (o & 3); the four k-lanes of an A fragment (o & 3 = lane >> 4) then
//     read four disjoint 8-bank groups.  Two tile buffers where they fit (C = 96): the next tile lands while this one is computed;
//   * output channels in passes of 48 (3 MFMA column tiles; a wave owns one row of 32 pixels = 2 row tiles, 6 accumulators): the x and
//     old-gradient values of a pass are requested BEFORE its K loop and used after it -- the HBM latency of the read-modify-write hides
//     under the pass's MFMAs instead of ending the block;
//   * the BN-backward sums run in per-wave LDS slots over the block's tiles (fp32: 8 pixels per tile and slot) and leave once, the waves
//     in a fixed order, one fp64 atomic per (channel, sum) and block.
// MFMA roles as everywhere: A[i = pixel x][k = o], B[k = o][j = c]; D lands with 4 consecutive pixels per lane for c = lane & 15.
#pragma once

#include "conv_dma_kernels.h"

namespace endo {

template <int C>
struct TdDgradGeom {
    static_assert(C % 48 == 0 && C % 16 == 0, "passes of 48 output channels");
    static constexpr int kThreads = 512;
    static constexpr int kTileX = 32, kTileY = 8;
    static constexpr int kWS = (C % 32 == 16) ? C : C + 16;            // weight row stride == 16 (mod 32) dwords
    static_assert(kWS % 32 == 16, "weight row stride");
    static constexpr int kTileFloats = C * 64 + C * 16;                // pooled gradient [C][16 units x 4] (swizzled) + codes [C][16 dwords]
    static constexpr int kBufs = (4 * (C * kWS + 2 * kTileFloats + 8 * C * 2 + 4 * C) <= 160 * 1024) ? 2 : 1;
    static constexpr int kFloats = C * kWS + kBufs * kTileFloats + 8 * C * 2 + 4 * C;          // weights, tiles, per-wave sums, BN constants
    static constexpr size_t kBytes = sizeof(float) * kFloats;
    static_assert(kBytes <= 160 * 1024, "one block per CU");
    static constexpr int kPasses = C / 48;
};

// p: the ConvParams of the per-tile launch (td_bwd): in / in_idx = pooled gradient and codes, wgt = W[o][c] (row stride w_cin), out / x /
// bn_* of the C output channels; p.cin == p.cout == C, p.w % 32 == 0, p.h % 8 == 0.  tiles_xy = tiles per sample, gn = samples per
// group, bpg = blocks per group (gridDim.x = bpg * groups).
template <int C>
__global__ void __launch_bounds__(512, 2) td_dgrad_kernel(const ConvParams p, int tiles_xy, int gn, int bpg) {
    using G = TdDgradGeom<C>;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int grp = blockIdx.x / bpg;
    const int r0 = blockIdx.x - grp * bpg;
    const int64_t grp_off = grp * p.gs;
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
    float* s_w = smem;                                   // [o][kWS]
    float* s_t = s_w + C * G::kWS;                       // [buf][ dY: C x 64 | codes: C x 16 ]
    float* s_sum = s_t + G::kBufs * G::kTileFloats;      // [wave 8][C][2]
    float* s_bn = s_sum + 8 * C * 2;                     // [C][scale, beta, mean, rstd]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = the wave's row of the tile
    const int li = lane & 15;
    const int lk = lane >> 4;

    // ---- once per block: weights, BN constants, zeroed sums ----
    for (int i = tid; i < C * C / 4; i += G::kThreads) {
        const int o = i / (C / 4), c4 = (i - o * (C / 4)) * 4;
        *reinterpret_cast<f32x4*>(s_w + o * G::kWS + c4) = *reinterpret_cast<const f32x4*>(p.wgt + static_cast<int64_t>(o) * p.w_cin + c4);
    }
    for (int c = tid; c < C; c += G::kThreads) {
        const float mean = p.bn_saved[grp_off + 2 * c], rstd = p.bn_saved[grp_off + 2 * c + 1];
        *reinterpret_cast<f32x4*>(s_bn + 4 * c) = f32x4{p.bn_gamma[c] * rstd, p.bn_beta[c], mean, rstd};
    }
    for (int i = tid; i < 8 * C * 2; i += G::kThreads) s_sum[i] = 0.f;
    if (t_begin >= t_end) return;          // (block-uniform)

    auto rsrc = [](const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7ffffffc, 0x00020000); };
    auto ld4 = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); };
    auto st4 = [](const f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0); };

    auto tile_origin = [&](int t, int& n, int& x0, int& y0) {
        n = t / tiles_xy;
        const int tile = t - n * tiles_xy;
        const int ty = tile / p.tiles_x;
        x0 = (tile - ty * p.tiles_x) * G::kTileX;
        y0 = ty * G::kTileY;
    };
    // ---- a tile's pooled gradient and codes -> buffer `buf`: 16-byte DMA, 4 channels (gradient) / 16 channels (codes) per instruction.
    //      Gradient: lane = (channel o = 4 j + (lane >> 4), LDS unit q = lane & 15), source unit q ^ 2 (o & 3) = (pooled row u >> 2, columns 4 (u & 3) ..)
    const int dq = (lane & 15) ^ (2 * (lane >> 4));
    const unsigned dy_lane = 4u * static_cast<unsigned>((lane >> 4) * p.in_cs + (dq >> 2) * p.in_w + 4 * (dq & 3));
    //      Codes: lane = (channel 16 j + (lane >> 2), pooled row lane & 3): the row's 16 code bytes
    const unsigned code_lane = static_cast<unsigned>((lane >> 2) * p.in_cs + (lane & 3) * p.in_w);
    auto issue_tile = [&](int t, int buf) {
        int n, x0, y0;
        tile_origin(t, n, x0, y0);
        const unsigned ppos = static_cast<unsigned>((y0 >> 1) * p.in_w + (x0 >> 1));
        const __amdgpu_buffer_rsrc_t gr = rsrc(p.in + grp_off + static_cast<int64_t>(n) * p.in_ns);
        const __amdgpu_buffer_rsrc_t cr = rsrc(p.in_idx + 4 * grp_off + static_cast<int64_t>(n) * p.idx_ns);
        float* dst = s_t + buf * G::kTileFloats;
        // the 8 waves share the C / 4 + C / 16 instructions
#pragma unroll
        for (int j0 = 0; j0 < (C / 4 + 7) / 8; ++j0) {
            const int j = j0 * 8 + wave;
            if (j < C / 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lptr_t)(dst + j * 256), 16, dy_lane + 4u * ppos, 4u * static_cast<unsigned>(4 * j * p.in_cs), 0, 0);
        }
#pragma unroll
        for (int j0 = 0; j0 < (C / 16 + 7) / 8; ++j0) {
            const int j = j0 * 8 + wave;
            if (j < C / 16)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(cr, (lptr_t)(dst + C * 64 + j * 256), 16, code_lane + ppos, static_cast<unsigned>(16 * j * p.in_cs), 0, 0);
        }
    };

    // ---- fragment addressing (per lane, fixed): A for row tile t in {0, 1} of the wave's row (pixels 16 t + li) ----
    const int py = wave >> 1;                             // pooled row of the wave's row
    const unsigned want = 2u * (wave & 1) + (li & 1);     // the argmax code that routes the pooled gradient to this lane's pixel
    const unsigned cshift = 8u * ((li >> 1) & 3);
    int a_off[2], c_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int unit = py * 4 + 2 * t + (li >> 3);
        a_off[t] = lk * 64 + ((unit ^ (2 * lk)) * 4) + ((li >> 1) & 3);          // + ks * 256
        c_off[t] = C * 64 + lk * 16 + unit;                                      // + ks * 64
    }
    const int b_off = lk * G::kWS + li;                  // + ks * 4 * kWS + pass * 48 + nt * 16

    // the lane's 4 pixels of (row tile t, column tile nt of a pass): channel li of the column tile, pixels 16 t + 4 lk ..
    const unsigned v_lane = 4u * static_cast<unsigned>(li * p.out_cs + wave * p.out_w + 4 * lk);          // (x_cs == out_cs, same planes)

    // x and old gradient of one pass of a tile: 2 row tiles x 3 column tiles, 16 bytes each
    f32x4 xv[2][3], ov[2][3], xn[2][3], on[2][3];
    auto load_pass = [&](__amdgpu_buffer_rsrc_t xr_, __amdgpu_buffer_rsrc_t or_, unsigned tb, int pass, f32x4 (&xd)[2][3], f32x4 (&od)[2][3]) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const unsigned so = tb + 4u * static_cast<unsigned>((pass * 48 + nt * 16) * p.out_cs);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                xd[tt][nt] = ld4(xr_, v_lane + 64u * tt, so);
                od[tt][nt] = ld4(or_, v_lane + 64u * tt, so);
            }
        }
    };
    int buf = 0;
    issue_tile(t_begin, 0);
    {
        int n, x0, y0;
        tile_origin(t_begin, n, x0, y0);
        load_pass(rsrc(p.x + grp_off + static_cast<int64_t>(n) * p.x_ns), rsrc(p.out + grp_off + static_cast<int64_t>(n) * p.out_ns),
                  4u * static_cast<unsigned>(y0 * p.out_w + x0), 0, xv, ov);
    }
    for (int t = t_begin; t < t_end; t += t_step) {
        int n, x0, y0;
        tile_origin(t, n, x0, y0);
        const float* x_n = p.x + grp_off + static_cast<int64_t>(n) * p.x_ns;
        float* out_n = p.out + grp_off + static_cast<int64_t>(n) * p.out_ns;
        const __amdgpu_buffer_rsrc_t xr = rsrc(x_n), orr = rsrc(out_n);
        const unsigned tile_b = 4u * static_cast<unsigned>(y0 * p.out_w + x0);
        const bool has_next = t + t_step < t_end;
        // This tile's maps have landed.  Two buffers: every wave has CONSUMED register loads younger than its DMA of this tile (the previous tile's
        // passes; loads retire in order), so only the first tile needs a wait; the barrier publishes the maps and retires the previous tile's
        // reads of the other buffer, which the next tile's DMA -- issued behind it -- then takes.  One buffer: the DMA went out behind the
        // previous tile's last K loop, nothing younger has been consumed: wait for it.
        if (G::kBufs == 1 || t == t_begin) __builtin_amdgcn_s_waitcnt(0x0070); else __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (G::kBufs == 2 && has_next) issue_tile(t + t_step, buf ^ 1);          // lands while this tile is computed
        const float* s_dy = s_t + buf * G::kTileFloats;

#pragma unroll 1
        for (int pass = 0; pass < G::kPasses; ++pass) {
            // ---- x and the old gradient of the NEXT pass (of the next tile's first pass): requested here, in flight over this pass's K
            //      loop and epilogue -- the read-modify-write's HBM reads never wait for an epilogue to finish ----
            {
                const bool last = pass == G::kPasses - 1;
                if (!last) load_pass(xr, orr, tile_b, pass + 1, xn, on);
                else if (has_next) {
                    int n2, x2, y2;
                    tile_origin(t + t_step, n2, x2, y2);
                    load_pass(rsrc(p.x + grp_off + static_cast<int64_t>(n2) * p.x_ns), rsrc(p.out + grp_off + static_cast<int64_t>(n2) * p.out_ns),
                              4u * static_cast<unsigned>(y2 * p.out_w + x2), 0, xn, on);
                }
            }
            f32x4 acc[2][3];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) acc[tt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* bw = s_w + b_off + pass * 48;
#pragma unroll 4
            for (int ks = 0; ks < C / 4; ++ks) {
                float a[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const float g = s_dy[a_off[tt] + ks * 256];
                    const unsigned cw = __float_as_uint(s_dy[c_off[tt] + ks * 64]);
                    a[tt] = ((cw >> cshift) & 0xffu) == want ? g : 0.f;
                }
                float b[3];
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) b[nt] = bw[ks * 4 * G::kWS + nt * 16];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) acc[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt], b[nt], acc[tt][nt], 0, 0, 0);
            }
            // single buffer: after the LAST pass's K loop the tile's maps are dead -- the next tile's DMA overlaps this epilogue
            if (G::kBufs == 1 && pass == G::kPasses - 1 && has_next) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                issue_tile(t + t_step, 0);
            }
            // ---- epilogue of the pass: ReLU mask, BN backward, read-modify-write, sums ----
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const int co = pass * 48 + nt * 16 + li;
                const f32x4 bn = *reinterpret_cast<const f32x4*>(s_bn + 4 * co);          // scale, beta, mean, rstd
                const bool accumulate = co >= p.acc_from;
                float s1 = 0.f, s2 = 0.f;
                const unsigned so = tile_b + 4u * static_cast<unsigned>((pass * 48 + nt * 16) * p.out_cs);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    f32x4 o = ov[tt][nt];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xc = xv[tt][nt][e] - bn[2];
                        const float z = fmaf(xc, bn[0], bn[1]);
                        const float dz = z > 0.f ? acc[tt][nt][e] : 0.f;
                        s1 += dz;
                        s2 += dz * (xc * bn[3]);
                        o[e] = (accumulate ? o[e] : 0.f) + bn[0] * dz;
                    }
                    // (offset all in the vector register: with a scalar offset register the compiler does not insert the wait state a 16-byte
                    // store needs before a VALU write of its data registers -- it overwrote the first dword of 0.1 % of these stores)
                    st4(o, orr, v_lane + 64u * tt + so, 0u);
                }
                // over the wave's four 16-lane rows (pixel groups), then the wave's slot.  (__shfl_xor, not the v_permlane swaps of
                // dgrad_wino3p_kernels.h: inline assembly right behind a 16-byte store is invisible to the compiler's hazard recogniser --
                // a build with them overwrote the first dword of 0.1 % of the stores; the two LDS round trips are per pass here, not per step)
                s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
                if (lk == 0) {          // the wave's slot runs over the block's whole run of tiles (owned: plain read-add-write, fp32; fp64 at the end)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2* slot = reinterpret_cast<f32x2*>(s_sum + (wave * C + co) * 2);
                    *slot = *slot + f32x2{s1, s2};
                }
            }
            // the next pass's operands become this pass's
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) { xv[tt][nt] = xn[tt][nt]; ov[tt][nt] = on[tt][nt]; }
        }
        buf ^= (G::kBufs == 2) ? 1 : 0;
    }
    // ---- once per block: the 8 waves' slots in a fixed order, one fp64 atomic per (channel, sum) ----
    __syncthreads();
    if (tid < 2 * C) {
        double v = 0.0;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) v += static_cast<double>(s_sum[wv * C * 2 + tid]);
        atomicAdd(p.bn_scratch + grp_off / 2 + bn_slot_offset(p.bn_slot_stride) + tid, v);
    }
}

inline bool td_dgrad_ok(const ConvParams& p) {
    return (p.cin == 96 || p.cin == 144) && p.cout == p.cin && p.w_cin == p.cin && (p.w % 32) == 0 && (p.h % 8) == 0 && (p.in_w % 4) == 0 &&
           (p.in_cs % 4) == 0 && (p.idx_ns % 16) == 0 && p.x_cs == p.out_cs && p.x_ns == p.out_ns && p.in_w == p.w / 2 && p.out_w == p.w &&
           (reinterpret_cast<uintptr_t>(p.in_idx) % 16) == 0 && (reinterpret_cast<uintptr_t>(p.wgt) % 16) == 0 && p.ksplit == 0 &&
           static_cast<int64_t>(p.cin) * p.out_cs * 4 < (1ll << 31);
}

template <int C>
inline int launch_td_dgrad_t(ConvParams p, int blocks, hipStream_t stream) {
    using G = TdDgradGeom<C>;
    p.tiles_x = p.w / G::kTileX;
    const int tiles_xy = p.tiles_x * (p.h / G::kTileY);
    const int groups = p.group_n > 0 ? p.n / p.group_n : 1;
    const int gn = p.group_n > 0 ? p.group_n : p.n;
    int bpg = blocks / groups;
    if (bpg >= 8) bpg &= ~7;
    if (bpg > tiles_xy * gn) bpg = tiles_xy * gn;
    if (bpg < 1) bpg = 1;
    ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(td_dgrad_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(G::kBytes)));
    td_dgrad_kernel<C><<<dim3(bpg * groups), G::kThreads, G::kBytes, stream>>>(p, tiles_xy, gn, bpg);
    ENDO_LAUNCH_CHECK();
    return 0;
}

inline int launch_td_dgrad(const ConvParams& p, int blocks, hipStream_t stream) {
    if (!td_dgrad_ok(p)) return ENDO_E_UNSUPPORTED;
    return p.cin == 96 ? launch_td_dgrad_t<96>(p, blocks, stream) : launch_td_dgrad_t<144>(p, blocks, stream);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same function at the coarse levels (64 x 80 ... 16 x 20 pixels, 192 ... 288 channels), where conv_dma_kernel<1, 16, 2, IN_UNPOOL>
// is a chain of 12 - 18 K-chunks per block that waits for dword LDS-DMA (140 / 63 us at levels 2 / 3 in the step) and, at level 4 -- the
// pooled rows are 10 bytes of codes: no whole dwords -- the register-staged conv_mfma_kernel takes 119 us for 0.85 GFLOP.
// Here a block owns 128 consecutive pixels of a sample's plane (any width that is a multiple of 4: rows are not tiles) x 48 output
// channels; a K-chunk of 16 pooled-gradient channels is EXPANDED on its way into LDS -- thread (pixel, channel) loads the pooled value and
// its argmax byte and stores the routed value U[o][p] -- so the K loop is plain fragment reads; loads of chunk k + 1 are in registers while
// chunk k is multiplied; x and the old gradient are requested before the K loop as in td_dgrad_kernel.
constexpr int kTdsPix = 128, kTdsN = 48, kTdsKC = 16;
constexpr int kTdsUS = kTdsPix + 16;          // U row stride == 16 (mod 32) dwords: the two k-lanes of a 32-lane pass read disjoint bank halves

__global__ void __launch_bounds__(256) td_dgrad_small_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float s_u[2][kTdsKC * kTdsUS];
    __shared__ __attribute__((aligned(16))) float s_w[2][kTdsKC * kTdsN];
    __shared__ float s_sum[4][kTdsN][2];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int plane = p.h * p.w;
    const int P0 = blockIdx.x * kTdsPix;
    const int c0 = blockIdx.y * kTdsN;
    const int grp = p.group_n > 0 ? blockIdx.z / p.group_n : 0;
    const int n = blockIdx.z - grp * p.group_n;
    const int64_t grp_off = grp * p.gs;
    const float* dy_n = p.in + grp_off + static_cast<int64_t>(n) * p.in_ns;
    const uint8_t* idx_n = p.in_idx + 4 * grp_off + static_cast<int64_t>(n) * p.idx_ns;
    const float* x_n = p.x + grp_off + static_cast<int64_t>(n) * p.x_ns;
    float* out_n = p.out + grp_off + static_cast<int64_t>(n) * p.out_ns;

    // ---- staging role: pixel P0 + (tid & 127), channels (tid >> 7) + 2 i of a chunk ----
    const int sp = P0 + (tid & 127);
    const bool sp_ok = sp < plane;
    const int sy = sp_ok ? sp / p.w : 0, sx = sp_ok ? sp - sy * p.w : 0;
    const int spp = (sy >> 1) * p.in_w + (sx >> 1);
    const unsigned swant = 2u * (sy & 1) + (sx & 1);
    float rg[8]; unsigned rc[8]; float rw[3];
    const int nchunks = p.cin / kTdsKC;
    auto fetch = [&](int chunk) {
        const int o0 = chunk * kTdsKC + (tid >> 7);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t off = static_cast<int64_t>(o0 + 2 * i) * p.in_cs + spp;
            rg[i] = sp_ok ? dy_n[off] : 0.f;
            rc[i] = sp_ok ? idx_n[off] : 255u;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = tid + 256 * i, o = e / kTdsN, c = e - o * kTdsN;
            rw[i] = p.wgt[static_cast<int64_t>(chunk * kTdsKC + o) * p.w_cin + c0 + c];
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s_u[buf][((tid >> 7) + 2 * i) * kTdsUS + (tid & 127)] = rc[i] == swant ? rg[i] : 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) s_w[buf][tid + 256 * i] = rw[i];
    };

    // ---- the lane's output pixels: row tile 2 wave + tt, pixels 4 lk .. 4 lk + 3; channels c0 + 16 nt + li ----
    f32x4 xv[2][3], ov[2][3];
    bool px_ok[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int pp = P0 + (2 * wave + tt) * 16 + 4 * lk;
        px_ok[tt] = pp < plane;          // (plane % 4 == 0: a quad is all in or all out)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int64_t o = static_cast<int64_t>(c0 + nt * 16 + li) * p.out_cs + pp;
            xv[tt][nt] = px_ok[tt] ? *reinterpret_cast<const f32x4*>(x_n + o) : f32x4{0.f, 0.f, 0.f, 0.f};
            ov[tt][nt] = px_ok[tt] ? *reinterpret_cast<const f32x4*>(out_n + o) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 acc[2][3];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) acc[tt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    stash(0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < nchunks) fetch(chunk + 1);
#pragma unroll
        for (int ks = 0; ks < kTdsKC / 4; ++ks) {
            float a[2], b[3];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) a[tt] = s_u[buf][(4 * ks + lk) * kTdsUS + (2 * wave + tt) * 16 + li];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) b[nt] = s_w[buf][(4 * ks + lk) * kTdsN + nt * 16 + li];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) acc[tt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt], b[nt], acc[tt][nt], 0, 0, 0);
        }
        if (chunk + 1 < nchunks) stash(buf ^ 1);          // (the other buffer: its readers passed the barrier that ended chunk - 1)
        __syncthreads();
    }
    // ---- epilogue: ReLU mask, BN backward, read-modify-write, sums (td_dgrad_kernel's, pixel quads outside the plane skipped) ----
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int co = c0 + nt * 16 + li;
        const float mean = p.bn_saved[grp_off + 2 * co], rstd = p.bn_saved[grp_off + 2 * co + 1];
        const float scale = p.bn_gamma[co] * rstd, beta = p.bn_beta[co];
        const bool accumulate = co >= p.acc_from;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            if (!px_ok[tt]) continue;
            f32x4 o = ov[tt][nt];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xc = xv[tt][nt][e] - mean;
                const float z = fmaf(xc, scale, beta);
                const float dz = z > 0.f ? acc[tt][nt][e] : 0.f;
                s1 += dz;
                s2 += dz * (xc * rstd);
                o[e] = (accumulate ? o[e] : 0.f) + scale * dz;
            }
            *reinterpret_cast<f32x4*>(out_n + static_cast<int64_t>(co) * p.out_cs + P0 + (2 * wave + tt) * 16 + 4 * lk) = o;
        }
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        if (lk == 0) { s_sum[wave][nt * 16 + li][0] = s1; s_sum[wave][nt * 16 + li][1] = s2; }
    }
    __syncthreads();
    if (tid < 2 * kTdsN) {
        const int j = tid >> 1, which = tid & 1;
        const double t = static_cast<double>(s_sum[0][j][which]) + static_cast<double>(s_sum[1][j][which]) + static_cast<double>(s_sum[2][j][which]) +
                         static_cast<double>(s_sum[3][j][which]);
        atomicAdd(p.bn_scratch + grp_off / 2 + bn_slot_offset(p.bn_slot_stride) + 2 * (c0 + j) + which, t);
    }
}

inline bool td_dgrad_small_ok(const ConvParams& p) {
    return p.cin == p.cout && p.w_cin == p.cin && (p.cin % kTdsN) == 0 && (p.w % 4) == 0 && (p.h % 2) == 0 && p.in_w == p.w / 2 && p.out_w == p.w &&
           p.out_cs == p.h * p.w && p.x_cs == p.out_cs && (p.out_cs % 4) == 0 && (p.out_ns % 4) == 0 && (p.x_ns % 4) == 0 && p.ksplit == 0 &&
           (reinterpret_cast<uintptr_t>(p.out) % 16) == 0 && (reinterpret_cast<uintptr_t>(p.x) % 16) == 0;
}

inline int launch_td_dgrad_small(const ConvParams& p, hipStream_t stream) {
    if (!td_dgrad_small_ok(p)) return ENDO_E_UNSUPPORTED;
    td_dgrad_small_kernel<<<dim3((p.h * p.w + kTdsPix - 1) / kTdsPix, p.cout / kTdsN, p.n), 256, 0, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

}  // namespace endo
