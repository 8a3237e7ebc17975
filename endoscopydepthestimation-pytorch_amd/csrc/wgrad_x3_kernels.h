// 3x3 weight gradient of the growth-12 dense layers with its fp32 products on the bf16 matrix cores (the three-term split of common.h).
//
//   dW[co][ci][ky][kx] = sum_p a[ci][p] * G[co][p - (ky-1, kx-1)]          (a = relu(bn(x)), zero outside; G the prepared output gradient)
//
// The GEMM of wgrad_nsplit_kernels.h (M = 108 (co, tap) rows -> 7 row groups, N = input channels dealt out to the 4 waves, K = pixels:
// one 32-pixel row segment per chunk) with v_mfma_f32_16x16x32_bf16: a lane's 8 k values are 8 CONSECUTIVE pixels.
//   * x never touches LDS (as before): a lane loads its own 2 x 16 bytes per channel group two chunks ahead, applies BN + ReLU and splits
//     the 8 values into (hi, mid, lo) in registers -- 44 VALU for a fragment that feeds 7 row groups x 6 MFMAs.
//   * G is split ONCE per chunk and block, on its way into LDS (round 4's first x3 kernel split every fragment it read: 7 fragments
//     per wave and chunk, the kernel was VALU-bound at 480 us against 574 us for fp32 at level 0).  A row group's fragment is G shifted by
//     (1 - kx) pixels, so 16-byte fragment reads need G at three alignments: the window is staged as 9 planes, one per tap = (column
//     shift kx, window row 2 - ky), each [4 column chunks of 8 pixels][12 maps] 16-byte slots, loaded from global memory at the shifted
//     address (a dword-aligned 16-byte load), split, and written as 8-byte halves.  Row m = 12 tap + co of the GEMM then reads slot
//     60 tap + 12 lk + co of each term: consecutive m -> consecutive slots (plane stride 60 = 12 mod 16), conflict-free ds_read_b128.
//   * two LDS buffers, one barrier per chunk: the next chunk's window is loaded to registers before the MFMAs of this chunk and
//     split + stored behind them -- the matrix pipe (126 MFMAs = 2016 cycles per wave and chunk at NG = 3) runs beside that VALU work.
//   * six products per fragment pair, smallest first, into the same fp32 accumulator; per-block partial sums and the fixed-order
//     reduce of the n-split kernel (its row order here: m = 12 tap + co).
#pragma once

#include <type_traits>

#include "wgrad_nsplit_kernels.h"

namespace endo {

constexpr int kX3Seg = 32;                              // pixels per chunk
constexpr int kX3Plane = 60;                            // 16-byte slots per tap plane: 4 x 12 used + 12 pad (60 = 12 mod 16)
constexpr int kX3Term = 9 * kX3Plane;                   // slots per term
constexpr int kX3BufSlots = 3 * kX3Term;                // per buffer (the zero rows 108..111 live in pad slots)
constexpr int kX3Units = 3 * 12 * 3 * 8;                // (kx, co, row, 4-pixel unit) float4 loads per chunk: 864
constexpr int kX3Rounds = (kX3Units + kConvThreads - 1) / kConvThreads;          // 4

// One block of 4 waves per CU, one wave per SIMD with the whole 512-register budget: every latency is covered by the wave's own
// software pipeline -- two resident blocks at 256 registers each spilled (acc 84 + two split x sets + loads in flight), and the spill
// traffic inside the chunk loop cost 2-3x (measured).  Per chunk c the instruction stream of a wave is seven row-group
// steps of 6 NG MFMAs, each followed by one slice of the vector work for chunk c + 1 (BN + ReLU + split of one x channel group, or
// split + store of one round of the G window), which the matrix pipe covers: bf16 MFMAs do not occupy the vector lanes.
template <int NG>
__global__ void __launch_bounds__(kConvThreads, 1) wgrad_x3_kernel(const WgradParams p, float* __restrict__ partial, int chunks_per_block) {
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[2 * kX3BufSlots * 16];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int groups_total = (p.cin + 15) / 16;
    const int per_pass = (groups_total + gridDim.y - 1) / gridDim.y;
    const int pass_g0 = blockIdx.y * per_pass;
    const int pass_n = min(per_pass, groups_total - pass_g0);
    const int rw = (wave + blockIdx.x) & 3;
    const int gq = pass_n >> 2, grem = pass_n & 3;
    const int ngw = gq + (rw < grem ? 1 : 0);                 // groups of this wave (<= NG); the others run on zeros
    const int group0 = pass_g0 + rw * gq + min(rw, grem);
    const int segs = (p.w + kX3Seg - 1) / kX3Seg;
    const int chunks_total = segs * p.h * p.n;
    const int c_begin = blockIdx.x * chunks_per_block;
    const int c_end = min(c_begin + chunks_per_block, chunks_total);

    float sc[NG], mn[NG], bt[NG];
    bool ch_ok[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) ch_ok[g] = g < ngw && 16 * (group0 + g) + li < p.cin;
    int cst_grp = -1;
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

    // fragment slot of row m = 16 mg + li = 12 tap + co, lane group lk (8 pixels = one 16-byte slot).  Rows 108..111 (lanes 12..15 of
    // the last row group) read zeros from pad slots chosen to continue the bank pattern: plane 1 + lk's pad starts at 12 + 12 lk (mod 16)
    int aslot[kNsMG];
#pragma unroll
    for (int mg = 0; mg < kNsMG; ++mg) {
        const int m = 16 * mg + li;
        const int tap = m / 12, co = m - tap * 12;
        aslot[mg] = ((m < 108) ? tap * kX3Plane + lk * 12 + co : (1 + lk) * kX3Plane + 48 + (li - 12)) * 16;
    }
    for (int e = tid; e < 2 * 3 * 4 * 4 * 4; e += kConvThreads) {          // the zero slots: [buffer][term][plane 1..4][pad slot 0..3][dword]
        const int dw = e & 3, ps = (e >> 2) & 3, pl = (e >> 4) & 3, rest = e >> 6;          // rest = buffer * 3 + term
        const int bufi = rest / 3, term = rest - bufi * 3;
        reinterpret_cast<unsigned*>(smem_raw)[((bufi * kX3BufSlots + term * kX3Term + (1 + pl) * kX3Plane + 48 + ps) << 2) + dw] = 0u;
    }

    f32x4 acc[NG][kNsMG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int m = 0; m < kNsMG; ++m) acc[g][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* pad_zero = g_pad_consts + 4;
    f32x4 xr[2][NG][2];          // raw x of the next two chunks (slot = chunk parity)
    unsigned xr_ok[2] = {0, 0};
    Bf16x8Split bq[2][NG];       // split relu(bn(x)) of the chunk being computed / the next one (index = chunk parity)

    // this thread's staging units: (kx, co, window row r, 4-pixel unit u) -> global offset inside the window, LDS byte offset of its
    // 8-byte half slot in the tap plane 3 (2 - r) + kx
    int s_goff[kX3Rounds], s_lds[kX3Rounds], s_meta[kX3Rounds];          // meta: r | kx << 2 | valid << 4 | u << 5
#pragma unroll
    for (int k = 0; k < kX3Rounds; ++k) {
        const int e = k * kConvThreads + tid;
        const bool valid = e < kX3Units;
        const int kx = valid ? e / 288 : 0;
        const int rem = e - kx * 288;
        const int co = valid ? rem / 24 : 0;
        const int r = valid ? (rem - co * 24) >> 3 : 0;
        const int u = rem & 7;
        s_goff[k] = co * p.dy_cs + r * p.dy_w + 4 * u + (1 - kx);
        // (units past the end of the last round write their zeros to a pad slot nobody reads: no predicate, no branch in the loop)
        s_lds[k] = valid ? (((3 * (2 - r) + kx) * kX3Plane + (u >> 1) * 12 + co) * 16) + (u & 1) * 8 : (6 * kX3Plane + 56) * 16 + (tid & 1) * 8;
        s_meta[k] = r | (kx << 2) | (valid ? 16 : 0) | (u << 5);
    }
    int x_off[NG];          // inside one sample: channels x plane fits 32 bits
#pragma unroll
    for (int g = 0; g < NG; ++g) x_off[g] = (16 * (group0 + g) + li) * p.in_cs + 8 * lk;

    int i_n = c_begin / (segs * p.h);
    int i_y = (c_begin - i_n * segs * p.h) / segs;
    int i_seg = c_begin - (i_n * p.h + i_y) * segs;
    int c_n = i_n, c_y = i_y, c_seg = i_seg;          // position of the next chunk whose x is transformed (BN constants of its sample group)
    int d_n = i_n, d_y = i_y, d_seg = i_seg;          // position of the next window to load

    f32x4 gw[kX3Rounds];
    unsigned gw_fix = 0;          // 2 bits per round: 1 = the unit hangs over the left image edge, 2 = over the right edge
    auto window_issue = [&]() {
        const int x0 = d_seg * kX3Seg;
        const WgSample sm(p, d_n);
        const float* base = p.dy + sm.dy_off(p) + static_cast<int64_t>(d_y - 1) * p.dy_w + x0;
        gw_fix = 0;
#pragma unroll
        for (int k = 0; k < kX3Rounds; ++k) {
            const int r = s_meta[k] & 3, kx = (s_meta[k] >> 2) & 3;
            const int col = x0 + 4 * (s_meta[k] >> 5);
            const bool ok = (s_meta[k] & 16) && static_cast<unsigned>(d_y - 1 + r) < static_cast<unsigned>(p.h) && col < p.w;
            const int dl = (kx == 2 && col == 0) ? 1 : 0;               // G[.][-1] is outside: load one to the right, shift in a zero
            const int dr = (kx == 0 && col + 4 == p.w) ? 1 : 0;         // G[.][w] is outside: load one to the left
            const float* src = ok ? base + s_goff[k] + dl - dr : pad_zero;
            gw[k] = *reinterpret_cast<const f32x4*>(src);
            gw_fix |= static_cast<unsigned>(ok ? dl + 2 * dr : 0) << (2 * k);
        }
        if (++d_seg == segs) {
            d_seg = 0;
            if (++d_y == p.h) { d_y = 0; ++d_n; }
        }
    };
    // one round of the window: registers -> split -> the three terms' planes of buffer `buf`
    auto window_store_round = [&](int buf, auto k_c) {
        constexpr int k = decltype(k_c)::value;
        unsigned char* s_buf = smem_raw + buf * kX3BufSlots * 16;
        const f32x4 w = gw[k];
        const unsigned fix = (gw_fix >> (2 * k)) & 3u;          // selects, not branches
        const f32x4 v = {fix == 1u ? 0.f : (fix == 2u ? w[1] : w[0]), fix == 1u ? w[0] : (fix == 2u ? w[2] : w[1]),
                         fix == 1u ? w[1] : (fix == 2u ? w[3] : w[2]), fix == 1u ? w[2] : (fix == 2u ? 0.f : w[3])};
        unsigned h0, m0, l0, h1, m1, l1;
        split_bf16x3_pair(v[0], v[1], h0, m0, l0);
        split_bf16x3_pair(v[2], v[3], h1, m1, l1);
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2_t*>(s_buf + s_lds[k]) = u32x2_t{h0, h1};
        *reinterpret_cast<u32x2_t*>(s_buf + kX3Term * 16 + s_lds[k]) = u32x2_t{m0, m1};
        *reinterpret_cast<u32x2_t*>(s_buf + 2 * kX3Term * 16 + s_lds[k]) = u32x2_t{l0, l1};
    };
    auto x_issue = [&](auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        const int x0 = i_seg * kX3Seg;
        const WgSample sm(p, i_n);
        const float* in_base = p.in + sm.in_off(p) + static_cast<int64_t>(i_y) * p.in_w + x0;
        unsigned okbits = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (x0 + 8 * lk + 4 * q < p.w) okbits |= 1u << q;
        xr_ok[slot] = okbits;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool ok = ch_ok[g] && (okbits & (1u << q));
                const float* src = ok ? in_base + x_off[g] + 4 * q : pad_zero;
                xr[slot][g][q] = *reinterpret_cast<const f32x4*>(src);
            }
        if (++i_seg == segs) {
            i_seg = 0;
            if (++i_y == p.h) { i_y = 0; ++i_n; }
        }
    };
    // BN + ReLU + split of channel group g of the raw x in `slot` -> bq[slot][g]
    auto x_transform = [&](auto slot_c, auto g_c) {
        constexpr int slot = decltype(slot_c)::value, g = decltype(g_c)::value;
        float v[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool ok = ch_ok[g] && (xr_ok[slot] & (1u << q));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * q + e] = ok ? __builtin_fmaxf(fmaf(xr[slot][g][q][e] - mn[g], sc[g], bt[g]), 0.f) : 0.f;
        }
        bq[slot][g] = split_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    };
    auto advance_consts = [&]() {          // the BN constants of the sample group of the next chunk to transform
        const int sg = WgSample(p, c_n).grp;
        if (sg != cst_grp) load_consts(sg);
        if (++c_seg == segs) {
            c_seg = 0;
            if (++c_y == p.h) { c_y = 0; ++c_n; }
        }
    };

    using Slot0 = std::integral_constant<int, 0>;
    using Slot1 = std::integral_constant<int, 1>;
    // prologue: window and x of the first chunk in place, x of the second in flight
    if (c_begin < c_end) {
        window_issue();
        x_issue(Slot0{});
        if (c_begin + 1 < c_end) x_issue(Slot1{});
        window_store_round(0, std::integral_constant<int, 0>{});
        window_store_round(0, std::integral_constant<int, 1>{});
        window_store_round(0, std::integral_constant<int, 2>{});
        window_store_round(0, std::integral_constant<int, 3>{});
        advance_consts();
        x_transform(Slot0{}, std::integral_constant<int, 0>{});
        if constexpr (NG > 1) x_transform(Slot0{}, std::integral_constant<int, 1>{});
        if constexpr (NG > 2) x_transform(Slot0{}, std::integral_constant<int, 2>{});
    }
    int buf = 0;
    auto do_chunk = [&](int chunk, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        using Next = std::integral_constant<int, slot ^ 1>;
        const bool more = chunk + 1 < c_end;
        __syncthreads();                                   // window `buf` is complete; everybody is done reading the other buffer
        if (more) {
            window_issue();                                // chunk + 1's window -> registers (stored behind row groups 3..6)
            advance_consts();
        }
        if (chunk + 2 < c_end) x_issue(slot_c);            // this chunk's raw x was consumed (split) during the previous chunk
        const unsigned char* s_buf = smem_raw + buf * kX3BufSlots * 16;
        bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(s_buf + aslot[0]);
        bf16x8_t am = *reinterpret_cast<const bf16x8_t*>(s_buf + kX3Term * 16 + aslot[0]);
        bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(s_buf + 2 * kX3Term * 16 + aslot[0]);
        auto step = [&](auto m_c) {
            constexpr int m = decltype(m_c)::value;
            bf16x8_t nh = ah, nm = am, nl = al;
            if constexpr (m + 1 < kNsMG) {
                nh = *reinterpret_cast<const bf16x8_t*>(s_buf + aslot[m + 1]);
                nm = *reinterpret_cast<const bf16x8_t*>(s_buf + kX3Term * 16 + aslot[m + 1]);
                nl = *reinterpret_cast<const bf16x8_t*>(s_buf + 2 * kX3Term * 16 + aslot[m + 1]);
            }
            // six products, smallest first; consecutive MFMAs go to different accumulators
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bq[slot][g].hi, acc[g][m], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bq[slot][g].lo, acc[g][m], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bq[slot][g].mid, acc[g][m], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bq[slot][g].hi, acc[g][m], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bq[slot][g].mid, acc[g][m], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bq[slot][g].hi, acc[g][m], 0, 0, 0);
            // this step's slice of the vector work for chunk + 1, in the same basic block (after the last chunk it runs on stale
            // registers into a buffer nobody reads): the group barriers below pin the order 1 MFMA, 4 VALU, 1 MFMA, ... -- a wave issues
            // in order, so only vector instructions placed BETWEEN two MFMAs run under the matrix pipe's 16 cycles
            if constexpr (m < NG) x_transform(Next{}, std::integral_constant<int, (m < NG ? m : 0)>{});
            else if constexpr (m - NG < kX3Rounds) window_store_round(buf ^ 1, std::integral_constant<int, (m >= NG && m - NG < kX3Rounds ? m - NG : 0)>{});
#pragma unroll
            for (int i = 0; i < 6 * NG; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);          // four VALU
            }
            __builtin_amdgcn_sched_barrier(0);          // live ranges end with the step
            ah = nh; am = nm; al = nl;
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{});
        buf ^= 1;
    };
    for (int chunk = c_begin; chunk < c_end; chunk += 2) {
        do_chunk(chunk, Slot0{});
        if (chunk + 1 < c_end) do_chunk(chunk + 1, Slot1{});
    }

    // partial[(group * 7 + m) * blocks + block][r][lane] = D[row 16 m + 4 lk + r][ci = 16 group + li] (wgrad_nsplit_kernels.h)
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

inline bool wgrad_x3_ok(const WgradParams& p) { return wgrad_nsplit_ok(p); }

template <int NG>
inline int launch_wgrad_x3_ng(const WgradParams& p, float* scratch, int passes, hipStream_t stream) {
    const int chunks_total = ((p.w + kX3Seg - 1) / kX3Seg) * p.h * p.n;
    int blocks = 256 / passes;                                  // one resident block per CU (one wave per SIMD with all the registers)
    if (blocks < 1) blocks = 1;
    const int per = (chunks_total + blocks - 1) / blocks;
    blocks = (chunks_total + per - 1) / per;
    const int groups_total = (p.cin + 15) / 16;
    wgrad_x3_kernel<NG><<<dim3(blocks, passes), kConvThreads, 0, stream>>>(p, scratch, per);
    ENDO_LAUNCH_CHECK();
    wgrad_nsplit_reduce_kernel<1><<<dim3(groups_total * kNsMG, 16), 256, 0, stream>>>(scratch, blocks, groups_total, p.cin, p.dw);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// scratch: kNsScratchFloats floats
inline int launch_wgrad_x3(const WgradParams& p, float* scratch, hipStream_t stream) {
    const int groups = (p.cin + 15) / 16;
    const int passes = (groups + 11) / 12;
    const int per_pass = (groups + passes - 1) / passes;
    const int ng = (per_pass + 3) / 4;
    if (ng <= 1) return launch_wgrad_x3_ng<1>(p, scratch, passes, stream);
    if (ng == 2) return launch_wgrad_x3_ng<2>(p, scratch, passes, stream);
    return launch_wgrad_x3_ng<3>(p, scratch, passes, stream);
}

}  // namespace endo
