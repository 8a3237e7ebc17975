// 3x3 weight gradient of the growth-12 dense layers in the Winograd domain, F(3x3, 4x4): a 4 x 4 tile of the output gradient G and the
// 6 x 6 activation patch around it give the tile's contribution to the 3 x 3 taps with 36 multiplications instead of 144.
//
//   dW[co][ci][ky][kx] = sum_p a[ci][p + (ky-1, kx-1)] * G[co][p]          (a = relu(bn(x)), zero outside the image)
//   per tile T (4 x 4 pixels) and (ci, co):   dW_T = A^T [ (S g S^T) .* (B^T d B) ] A,   g = G[co] on T (4 x 4),  d = a[ci] on T's 6 x 6 patch
//   (points 0, +-1, +-2, inf: B^T the 6 x 6 input transform of F(4,3); S the 6 x 4 transform of the 4-tap "filter" g; A^T 3 x 6)
// summed over tiles BEFORE the output transform: 36 GEMMs M[xi][co][ci] = sum_tiles U[xi][co][tile] * X[xi][tile][ci] on the fp32 matrix
// cores (v_mfma_f32_16x16x4_f32: 16 co rows (12 used) x 16 ci columns, k = 4 tiles), one output transform per wave at the end.
// Against the direct n-split kernel (wgrad_nsplit_kernels.h: 28 MFMAs per 16 pixels and 16 channels) that is 9 MFMAs: fp32 MFMAs run on
// the vector lanes of this part, so matrix and vector instruction counts are both time.
// fp32 accuracy: 5e-6 (max) / 8e-7 (rms) of max |dW| against fp64 (tools/x3_bench), the direct kernel's 5e-7 / 1.5e-7.
//
// Work: every WAVE is on its own (no block-level sharing, no barriers): it owns one 16-channel group and a list of column segments
// (sample, rows, 16-column strip) and walks each downwards in steps of 4 rows; a step is 4 horizontally adjacent tiles (the k of an
// MFMA) x 36 xi, accumulated in 144 registers.  Two waves per SIMD (256 registers each) cover each other's latencies.
//   * x: lane (ci = lane & 15, tile = lane >> 4) needs the four new rows of ITS patch per step (rows 4 ty + 3, 4 ty + 4 of the last step
//     stay in registers).  Loading them lane by lane costs the L1 one tag look-up per lane and instruction (16 channel planes: no two
//     lanes share a line; the texture cache was 2/3 busy, the waves waited on it 38 % of their time), so they arrive COALESCED through
//     LDS-DMA: buffer_load_dwordx4 ... lds, lane e of instruction i = channel 2 i + (e >> 5), row (e >> 3) & 3, 16-byte unit e & 7 of a
//     32-column window, straight into one of two wave-private LDS images (8.4 KB each), two steps ahead, no registers; each lane then reads a float4 and
//     the two halo columns per row, applies BN + ReLU and runs the 6 x 6 input transform (packed fp32 where the pairs fall out naturally).
//   * G: lane (co = lane & 15, tile = lane >> 4) loads its 4 x 4 tile and transforms it in registers -- exactly the A operand's layout
//     (lanes co >= 12 load out of the descriptor's range, i.e. zeros).  The scale factors of S (1/4, -1/6, 1/24) are folded into the
//     output transform.
//   * XCD-aware order: the hardware deals blocks out to the 8 XCDs round-robin; XCD x owns the segments [x, x + 1) * units / 8, the waves
//     of a channel group take them interleaved, so that at any time they work on neighbouring strips of the same rows: the other half of
//     a strip's 128-byte lines and its halo columns are then in that XCD's L2 (with contiguous ranges per wave every line came from HBM
//     2.5 times, L2 hit rate 2 %).
//   * per-wave partial sums of the 9 taps (the output transform is linear, so it is applied to the wave's own sums) go to scratch;
//     wgrad_f34_reduce_kernel adds them in a fixed order and accumulates into the flat gradient.
// Needs w % 4 == 0 and h % 16 == 0 (levels 0-3 of the 256 x 320 configuration, 0-4 of 512 x 640).
#pragma once

#include <algorithm>

#include "conv_dma_kernels.h"
#include "wgrad_kernels.h"

namespace endo {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kF34XsPair = 264;                    // x staging image of a wave: 8 channel pairs x ([2 ci][4 rows][32 columns] + 8)
constexpr int kF34Xs = 8 * kF34XsPair;             // floats per wave (8448 bytes)
constexpr int kF34WavesPerXcd = 256;               // 32 CUs x 2 blocks x 4 waves

struct F34Plan {
    int groups;          // 16-channel groups
    int wpg;             // waves per group and XCD (>= 4 or the only ones): waves [g * wpg, (g + 1) * wpg) of an XCD work on group g
    int segq;            // quads (16 rows) per column segment, the unit of work
    int spb;             // partial-sum rows per (group, XCD): one per block that holds waves of the group (the largest count over the groups)
    int slots;           // partial-sum rows per (group, tap): 8 * spb (the rows past a group's last block are never written nor read)
};
// blocks (of four waves) of an XCD that hold waves of group g: first, count
__host__ __device__ inline int f34_first_block(const F34Plan& plan, int g) { return (g * plan.wpg) >> 2; }
__host__ __device__ inline int f34_block_count(const F34Plan& plan, int g) { return (((g + 1) * plan.wpg - 1) >> 2) - ((g * plan.wpg) >> 2) + 1; }
// is row `slot` (of the 8 * spb rows of a (group, tap)) one that a block wrote?
__host__ __device__ inline bool f34_row_written(const F34Plan& plan, int g, int slot) { return slot % plan.spb < f34_block_count(plan, g); }

// 6-point input transform B^T d (12 instructions; T = float, or f32x2 for two columns at once: v_pk_fma_f32 / v_pk_add_f32)
template <typename T>
__device__ __forceinline__ T f34_fma(const float a, const T b, const T c) {
    if constexpr (sizeof(T) == 8) return __builtin_elementwise_fma(T{a, a}, b, c);
    else return fmaf(a, b, c);
}
template <typename T>
__device__ __forceinline__ void f34_bt(const T d0, const T d1, const T d2, const T d3, const T d4, const T d5, T (&t)[6]) {
    t[0] = f34_fma<T>(4.f, d0, f34_fma<T>(-5.f, d2, d4));
    const T pp = f34_fma<T>(-4.f, d2, d4), qq = f34_fma<T>(-4.f, d1, d3);
    t[1] = pp + qq;
    t[2] = pp - qq;
    const T rr = d4 - d2, ss = d3 - d1;
    t[3] = f34_fma<T>(2.f, ss, rr);
    t[4] = f34_fma<T>(-2.f, ss, rr);
    t[5] = f34_fma<T>(4.f, d1, f34_fma<T>(-5.f, d3, d5));
}

// 4 -> 6 transform of the G tile WITHOUT the row scales (1/4, -1/6, -1/6, 1/24, 1/24, 1), 8 instructions
template <typename T>
__device__ __forceinline__ void f34_s(const T g0, const T g1, const T g2, const T g3, T (&u)[6]) {
    const T e = g0 + g2, o = g1 + g3;
    const T e4 = f34_fma<T>(4.f, g2, g0), o4 = f34_fma<T>(4.f, g3, g1);
    u[0] = g0;
    u[1] = e + o;
    u[2] = e - o;
    u[3] = f34_fma<T>(2.f, o4, e4);
    u[4] = f34_fma<T>(-2.f, o4, e4);
    u[5] = g3;
}

// Keeps a packed result in a register PAIR: without it the compiler splits v_pk_fma_f32 / v_pk_add_f32 into two scalar instructions
// wherever the halves are used one at a time (an empty asm: no instruction, no hazard the compiler does not see)
__device__ __forceinline__ void f34_pin(f32x2& v) { asm("" : "+v"(v)); }

// raw buffer descriptor (stride 0): loads are base + voffset (VGPR) + soffset (SGPR) + imm in ONE instruction, and an offset at or past
// `bytes` reads zeros without touching memory
__device__ __forceinline__ __amdgpu_buffer_rsrc_t f34_rsrc(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 f34_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float f34_ld1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// EXP (tools/x3_bench only): 2 = no x loads
// RAW: the convolution reads its input as it is (the network's first convolution: the image, 3 channels, no BatchNorm, no ReLU) and has
// 12 * plan.groups output channels: a wave's "group" is then a SET OF 12 OUTPUT CHANNELS (its 12 gradient planes), every wave transforms the
// same (<= 16) input channels
// PREP (with RAW): `dy` is the raw gradient and the kernel prepares it (WgradParams::prep_x): four more 16-byte loads per step, counted in the waits
template <int EXP = 0, bool RAW = false, bool PREP = false>
__global__ void __launch_bounds__(kConvThreads, 2) wgrad_f34_kernel(const WgradParams p, float* __restrict__ partial, const F34Plan plan) {
    extern __shared__ __attribute__((aligned(16))) float smem[];          // [wave][2 images]: the rows of the next two steps
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;

    // this wave's channel group and its place among the group's waves on this XCD
    // (the waves of a block belong to at most two groups: their sums meet in LDS at the end and leave as one row per group)
    const int xcd = blockIdx.x & 7;
    const int bi = blockIdx.x >> 3;
    const int wi = bi * 4 + wave;
    const bool active = wi < plan.groups * plan.wpg;
    const int group = active ? wi / plan.wpg : 0;
    const int wq = wi - group * plan.wpg;

    // work units: column segments (sample n, segment of `segq` quads of 16 rows, strip s), numbered with s fastest
    const int S = (p.w + 15) >> 4, CY = p.h >> 4, segq = plan.segq, YS = CY / segq;          // the last strip may be partly outside (w % 4 == 0)
    const int units = p.n * YS * S;
    const int per_xcd = (units + 7) >> 3;
    const int u_end = min(units, (xcd + 1) * per_xcd);
    const int u_first = active ? xcd * per_xcd + wq : u_end;

    f32x4 acc[36];
#pragma unroll
    for (int m = 0; m < 36; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // x side
    const int ch = RAW ? li : 16 * group + li;
    const bool ch_ok = ch < p.cin;
    const unsigned bo = 4u * static_cast<unsigned>(min(ch, p.cin - 1) * p.in_cs + 4 * lk);          // (lane-by-lane loads of a segment's first rows)
    float* const xs = smem + wave * 2 * kF34Xs;
    const unsigned x_vo = 4u * static_cast<unsigned>((lane >> 5) * p.in_cs + ((lane >> 3) & 3) * p.in_w + 4 * (lane & 7));
    const float* const xs_r = xs + (li >> 1) * kF34XsPair + (li & 1) * 128 + 4 * lk;
    const unsigned x_grp = RAW ? 0u : 4u * static_cast<unsigned>(16 * group * p.in_cs);
    // G side: the lane's (co, tile) offset; lanes co >= 12 point past the descriptor's range (zeros)
    const unsigned g_vo_in = li < 12 ? 4u * static_cast<unsigned>(li * p.dy_cs + 4 * lk) : 0x80000000u;

    float sc = 0.f, sh = 0.f;          // BN + ReLU as relu(sc * x + sh), constants of the current sample group
    int cst_grp = -1;
    float prep_pc = 0.f, prep_qc = 0.f, prep_sum = 0.f;          // RAW with p.prep_x: G = d + P x + Q formed here; sum G of this lane's tiles
    float keep[2][6];                  // activated image rows 4 ty - 1, 4 ty of the step to come, columns in the order (0, 5, 1, 2, 3, 4)

    // BN + ReLU of one row in the column order (0, 5 | 1, 2 | 3, 4): three packed fmas, six max
    auto act_row = [&](const float hl, const f32x4 m, const float hr, const f32x2 sc_e, const f32x2 sh_e, float (&o)[6]) {
        const f32x2 e = __builtin_elementwise_fma(f32x2{hl, hr}, sc_e, sh_e);
        const f32x2 a = __builtin_elementwise_fma(f32x2{m[0], m[1]}, f32x2{sc, sc}, f32x2{sh, sh});
        const f32x2 b = __builtin_elementwise_fma(f32x2{m[2], m[3]}, f32x2{sc, sc}, f32x2{sh, sh});
        if constexpr (RAW) {          // sc = 1 (0 for a masked column or channel), sh = 0; no max() here to turn "0 * garbage" into 0: the halo
            // column outside the image is read from LDS that was never written and is selected away
            o[0] = sc_e[0] != 0.f ? e[0] : 0.f; o[1] = sc_e[1] != 0.f ? e[1] : 0.f;
            o[2] = a[0]; o[3] = a[1]; o[4] = b[0]; o[5] = b[1];
        } else {
            o[0] = fmaxf(e[0], 0.f); o[1] = fmaxf(e[1], 0.f);
            o[2] = fmaxf(a[0], 0.f); o[3] = fmaxf(a[1], 0.f);
            o[4] = fmaxf(b[0], 0.f); o[5] = fmaxf(b[1], 0.f);
        }
    };
    // the four new rows of a step (image rows `row` .. `row` + 3) into the LDS image: 8 DMA instructions, no registers.  The window
    // starts 8 columns left of the strip (at column 0 for the first strip); rows below the image and columns past the row end are whatever
    // follows in memory (zeros past the sample) and are zeroed after BN + ReLU.
    // Of the window's 8 units the reads touch columns 7 .. 24 (units 1 .. 6), 0 .. 16 for the first strip (units 0 .. 4): the other lanes stay out.
    auto x_issue = [&](const __amdgpu_buffer_rsrc_t r, int s, int row, int img) {
        const unsigned so = x_grp + 4u * static_cast<unsigned>(row * p.in_w + 16 * s - (s > 0 ? 8 : 0));
        const int unit = lane & 7;
        float* dst = xs + img * kF34Xs;
        if (s > 0 ? (unit >= 1 && unit <= 6) : unit <= 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (!(EXP & 2))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)(dst + i * kF34XsPair), 16, x_vo, so + 8u * static_cast<unsigned>(i * p.in_cs), 0, 0);
        }
    };

    for (int u = u_first; u < u_end; u += plan.wpg) {
        const int s = u % S;
        const int rest = u / S;
        const int yseg = rest % YS;
        const int n = rest / YS;
        const WgSample sm(p, n);
        // descriptors over this sample's cin activation planes / 12 gradient planes: anything past them (the padding channels of the last
        // group, the rows below the last plane) reads zeros
        const __amdgpu_buffer_rsrc_t xr = f34_rsrc(p.in + sm.in_off(p), p.cin * p.in_cs * 4);
        const __amdgpu_buffer_rsrc_t gr = f34_rsrc(p.dy + sm.dy_off(p) + (RAW ? static_cast<int64_t>(12 * group) * p.dy_cs : 0), 12 * p.dy_cs * 4);
        const __amdgpu_buffer_rsrc_t pr = f34_rsrc((PREP ? p.prep_x : p.dy) + sm.dy_off(p) + (RAW ? static_cast<int64_t>(12 * group) * p.dy_cs : 0), 12 * p.dy_cs * 4);
        if (PREP && sm.grp != cst_grp) {
            prep_pc = li < 12 ? p.prep_p[sm.grp * p.gs + 12 * group + li] : 0.f;
            prep_qc = li < 12 ? p.prep_q[sm.grp * p.gs + 12 * group + li] : 0.f;
        }
        if (sm.grp != cst_grp) {
            sc = 0.f; sh = 0.f;
            if constexpr (RAW) {
                if (ch_ok) sc = 1.f;
            } else if (ch_ok) {
                const float* saved = p.saved + sm.grp * p.gs;
                sc = p.gamma[ch] * saved[2 * ch + 1];
                sh = fmaf(-saved[2 * ch], sc, p.beta[ch]);
            }
            cst_grp = sm.grp;
        }
        // halo columns outside the image: masked through the constants (relu(0 * x + 0) = 0)
        // a tile right of the image (last strip of a width that is no multiple of 16) takes part with a zero gradient tile
        const unsigned g_vo = 16 * s + 4 * lk < p.w ? g_vo_in : 0x80000000u;
        const bool l_out = s == 0 && lk == 0, r_out = 16 * s + 4 * lk + 4 >= p.w;
        const f32x2 sc_e = {l_out ? 0.f : sc, r_out ? 0.f : sc}, sh_e = {l_out ? 0.f : sh, r_out ? 0.f : sh};
        const int rd_shift = s > 0 ? 8 : 0;          // where the strip's first column sits in the staged window
        const int row_begin = 16 * yseg * segq, row_end = row_begin + 16 * segq;

        // the segment's first two steps: their new rows on their way (image 0, image 1), image rows row_begin - 1 and row_begin lane by
        // lane meanwhile (the previous segment's last steps have consumed both images: every LDS read of them was waited for)
        x_issue(xr, s, row_begin + 1, 0);
        if (row_begin + 4 < row_end) x_issue(xr, s, row_begin + 5, 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = row_begin - 1 + i;
            if (row >= 0) {
                const unsigned off = bo + 4u * static_cast<unsigned>(row * p.in_w + 16 * s);
                const f32x4 m = f34_ld4(xr, off, 0);
                const float a = f34_ld1(xr, off - (l_out ? 0u : 4u), 0), b = f34_ld1(xr, off + (r_out ? 0u : 16u), 0);
                act_row(a, m, b, sc_e, sh_e, keep[i]);
            } else {
#pragma unroll
                for (int e = 0; e < 6; ++e) keep[i][e] = 0.f;
            }
        }

#pragma unroll 1
        for (int row = row_begin; row < row_end; row += 4) {          // one step: output rows row .. row + 3
            // G tile of the step (zeros for co >= 12)
            f32x4 gt[4], xa[4];
            {
                const unsigned so = 4u * static_cast<unsigned>(row * p.dy_w + 16 * s);
                const unsigned pitch = 4u * static_cast<unsigned>(p.dy_w);
#pragma unroll
                for (int i = 0; i < 4; ++i) gt[i] = f34_ld4(gr, g_vo, so + i * pitch);
                if constexpr (PREP) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) xa[i] = f34_ld4(pr, g_vo, so + i * pitch);
                }
            }
            // the step's x rows are in LDS once at most the next step's 8 DMAs and the four (PREP: eight) G loads are in flight
            const int img = ((row - row_begin) >> 2) & 1;
            if (row + 4 < row_end) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PREP ? 16 : 12) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PREP ? 8 : 4) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            float d[6][6];          // [row][column in the order 0, 5, 1, 2, 3, 4]
#pragma unroll
            for (int e = 0; e < 6; ++e) { d[0][e] = keep[0][e]; d[1][e] = keep[1][e]; }
            {
                const float* rp = xs_r + img * kF34Xs + rd_shift;
                f32x4 m[4];
                float hl[4], hr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    m[i] = *reinterpret_cast<const f32x4*>(rp + 32 * i);
                    hl[i] = rp[32 * i - 1];
                    hr[i] = rp[32 * i + 4];
                }
                // every read of the image has returned: the next step's rows may overwrite it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (row + 8 < row_end) x_issue(xr, s, row + 9, img);          // two steps ahead, into the image just read
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) act_row(hl[i], m[i], hr[i], sc_e, sh_e, d[2 + i]);
            }
            if (row + 4 >= p.h) {          // (uniform, last step of a column) the row below the image
#pragma unroll
                for (int e = 0; e < 6; ++e) d[5][e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 6; ++e) { keep[0][e] = d[4][e]; keep[1][e] = d[5][e]; }

            if constexpr (PREP) {
                // G = d + P x + Q; a tile right of the image / a lane co >= 12 has read zeros for d and x and must stay zero: no Q there
                const float pc = g_vo != 0x80000000u ? prep_pc : 0.f, qc = g_vo != 0x80000000u ? prep_qc : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { gt[i][k] += fmaf(pc, xa[i][k], qc); prep_sum += gt[i][k]; }
            }
            // G: column pass on the column pairs (0, 1), (2, 3) of the tile
            f32x2 va[6], vb[6];
            f34_s(f32x2{gt[0][0], gt[0][1]}, f32x2{gt[1][0], gt[1][1]}, f32x2{gt[2][0], gt[2][1]}, f32x2{gt[3][0], gt[3][1]}, va);
            f34_s(f32x2{gt[0][2], gt[0][3]}, f32x2{gt[1][2], gt[1][3]}, f32x2{gt[2][2], gt[2][3]}, f32x2{gt[3][2], gt[3][3]}, vb);
            // x: column pass on the column pairs (0, 5), (1, 2), (3, 4)
            f32x2 t05[6], t12[6], t34[6];
            f34_bt(f32x2{d[0][0], d[0][1]}, f32x2{d[1][0], d[1][1]}, f32x2{d[2][0], d[2][1]}, f32x2{d[3][0], d[3][1]}, f32x2{d[4][0], d[4][1]},
                   f32x2{d[5][0], d[5][1]}, t05);
            f34_bt(f32x2{d[0][2], d[0][3]}, f32x2{d[1][2], d[1][3]}, f32x2{d[2][2], d[2][3]}, f32x2{d[3][2], d[3][3]}, f32x2{d[4][2], d[4][3]},
                   f32x2{d[5][2], d[5][3]}, t12);
            f34_bt(f32x2{d[0][4], d[0][5]}, f32x2{d[1][4], d[1][5]}, f32x2{d[2][4], d[2][5]}, f32x2{d[3][4], d[3][5]}, f32x2{d[4][4], d[4][5]},
                   f32x2{d[5][4], d[5][5]}, t34);
#pragma unroll
            for (int i = 0; i < 6; ++i) { f34_pin(va[i]); f34_pin(vb[i]); f34_pin(t05[i]); f34_pin(t12[i]); f34_pin(t34[i]); }
            // row passes of row i of both operands, then its 6 MFMAs
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                f32x2 eo = va[i] + vb[i], eo4 = f34_fma<f32x2>(4.f, vb[i], va[i]);
                f34_pin(eo); f34_pin(eo4);
                const float u0 = va[i][0], u1 = eo[0] + eo[1], u2 = eo[0] - eo[1];
                const float u3 = fmaf(2.f, eo4[1], eo4[0]), u4 = fmaf(-2.f, eo4[1], eo4[0]), u5 = vb[i][1];
                const f32x2 p05 = t05[i], p12 = t12[i], p34 = t34[i];
                const float x0 = fmaf(4.f, p05[0], fmaf(-5.f, p12[1], p34[1]));
                const float x5 = fmaf(4.f, p12[0], fmaf(-5.f, p34[0], p05[1]));
                f32x2 qp = f34_fma<f32x2>(-4.f, p12, p34);          // (T3 - 4 T1, T4 - 4 T2)
                f32x2 sr = p34 - p12;                               // (T3 - T1, T4 - T2)
                f34_pin(qp); f34_pin(sr);
                const float x1 = qp[1] + qp[0], x2 = qp[1] - qp[0];
                const float x3 = fmaf(2.f, sr[0], sr[1]), x4 = fmaf(-2.f, sr[0], sr[1]);
                acc[6 * i + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u0, x0, acc[6 * i + 0], 0, 0, 0);
                acc[6 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u1, x1, acc[6 * i + 1], 0, 0, 0);
                acc[6 * i + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(u2, x2, acc[6 * i + 2], 0, 0, 0);
                acc[6 * i + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(u3, x3, acc[6 * i + 3], 0, 0, 0);
                acc[6 * i + 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(u4, x4, acc[6 * i + 4], 0, 0, 0);
                acc[6 * i + 5] = __builtin_amdgcn_mfma_f32_16x16x4f32(u5, x5, acc[6 * i + 5], 0, 0, 0);
            }
        }
    }

    if constexpr (PREP) {
        if (p.prep_bias) {          // bias gradient: sum G over every tile this wave prepared (each tile of a channel is prepared by exactly one wave)
            float v = prep_sum;
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (active && lk == 0 && li < 12) atomicAdd(p.prep_bias + 12 * group + li, v);
        }
    }
    // output transform with the scales of S folded in: out[a][b] = sum_ij C[a][i] M[i][j] C[b][j]
    const float C[3][6] = {{0.25f, -1.f / 6.f, -1.f / 6.f, 1.f / 24.f, 1.f / 24.f, 0.f},
                           {0.f, -1.f / 6.f, 1.f / 6.f, 1.f / 12.f, -1.f / 12.f, 0.f},
                           {0.f, -1.f / 6.f, -1.f / 6.f, 1.f / 6.f, 1.f / 6.f, 1.f}};
    f32x4 h[3][6];          // rows combined: h[a][j] = sum_i C[a][i] M[i][j]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const f32x4 m1 = acc[6 + j], m2 = acc[12 + j], m3 = acc[18 + j], m4 = acc[24 + j];
        const f32x4 s12 = m1 + m2, d12 = m2 - m1, s34 = m3 + m4, d34 = m3 - m4;
        h[0][j] = 0.25f * acc[j] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
        h[1][j] = (1.f / 6.f) * d12 + (1.f / 12.f) * d34;
        h[2][j] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + acc[30 + j];
    }
    // the block's waves are done with their images: the 9 x 256 sums of each wave meet in LDS, one row per group of the block leaves
    __syncthreads();
    float* const red = smem + wave * (9 * 256);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (C[b][j] != 0.f) o += C[b][j] * h[a][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(3 * a + b) * 256 + r * 64 + lane] = o[r];
        }
    __syncthreads();
    const int total = plan.groups * plan.wpg;
    int gw[5];                 // group of each wave of the block (-1: none)
    int64_t row[4];            // where the sum that ends with wave w goes: the row of (group, tap 0) that belongs to this block
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        gw[w] = bi * 4 + w < total ? (bi * 4 + w) / plan.wpg : -1;
        row[w] = ((static_cast<int64_t>(max(gw[w], 0)) * 9) * plan.slots + xcd * plan.spb + (bi - f34_first_block(plan, max(gw[w], 0)))) * 256 + tid;
    }
    gw[4] = -1;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (gw[w] < 0) continue;
            sum += smem[w * (9 * 256) + t * 256 + tid];
            if (gw[w + 1] != gw[w]) {
                partial[row[w] + static_cast<int64_t>(t) * plan.slots * 256] = sum;
                sum = 0.f;
            }
        }
    }
}

// grid (groups * 9, slices): adds the partial rows of (group, tap) over the group's waves in a fixed order per slice
// raw: the groups are sets of 12 output channels over the same <= 16 input channels (wgrad_f34_kernel<.., RAW>)
__global__ void __launch_bounds__(256) wgrad_f34_reduce_kernel(const float* __restrict__ partial, const F34Plan plan, int cin, float* __restrict__ dw, int raw) {
    __shared__ f32x4 s_part[4][64];
    const int gm = blockIdx.x;
    const int group = gm / 9, tap = gm - group * 9;
    const int slots = plan.slots;
    const int per = (slots + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(slots, b0 + per);
    if (b0 >= b1) return;
    const int q = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const f32x4* src = reinterpret_cast<const f32x4*>(partial + (static_cast<int64_t>(gm) * slots) * 256) + q;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
    for (int b = b0 + sub; b < b1; b += 4)
        if (f34_row_written(plan, group, b)) s0 += src[static_cast<int64_t>(b) * 64];
    s_part[sub][q] = s0;
    __syncthreads();
    const int e = threadIdx.x;
    const float* sp = reinterpret_cast<const float*>(s_part);
    const float total = (sp[e] + sp[256 + e]) + (sp[512 + e] + sp[768 + e]);
    const int lane = e & 63, r = e >> 6;
    const int co = 4 * (lane >> 4) + r;
    const int ci = (raw ? 0 : 16 * group) + (lane & 15);
    if (co < 12 && ci < cin) atomicAdd(dw + (static_cast<int64_t>(co + (raw ? 12 * group : 0)) * cin + ci) * 9 + tap, total);
}

// The same for up to four launches at once (the four layers of a dense block, each with its own scratch slice): grid (max groups * 9, slices, layers).
// One launch instead of four 6-15 us ones whose time is mostly ramp-up and drain.
struct F34ReduceBatch {
    const float* partial[4];
    float* dw[4];
    F34Plan plan[4];
    int cin[4];
    int count;
};
__global__ void __launch_bounds__(256) wgrad_f34_reduce_batch_kernel(const F34ReduceBatch a) {
    __shared__ f32x4 s_part[4][64];
    const int l = blockIdx.z;
    const int gm = blockIdx.x;
    const F34Plan plan = a.plan[l];
    if (gm >= plan.groups * 9) return;          // (block-uniform)
    const int slots = plan.slots, cin = a.cin[l];
    const int group = gm / 9, tap = gm - group * 9;
    const int per = (slots + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(slots, b0 + per);
    if (b0 >= b1) return;
    const int q = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const f32x4* src = reinterpret_cast<const f32x4*>(a.partial[l] + (static_cast<int64_t>(gm) * slots) * 256) + q;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
    for (int b = b0 + sub; b < b1; b += 4)
        if (f34_row_written(plan, group, b)) s0 += src[static_cast<int64_t>(b) * 64];
    s_part[sub][q] = s0;
    __syncthreads();
    const int e = threadIdx.x;
    const float* sp = reinterpret_cast<const float*>(s_part);
    const float total = (sp[e] + sp[256 + e]) + (sp[512 + e] + sp[768 + e]);
    const int lane = e & 63, r = e >> 6;
    const int co = 4 * (lane >> 4) + r;
    const int ci = 16 * group + (lane & 15);
    if (co < 12 && ci < cin) atomicAdd(a.dw[l] + (static_cast<int64_t>(co) * cin + ci) * 9 + tap, total);
}

constexpr int kF34Blocks = 512;                      // two blocks of four waves per CU
constexpr int kF34MinTiles = 256;            // 4 x 4 tiles per launch from which the kernel is chosen (a quad of 16 x 16 pixels = 16 tiles)
// scratch: groups * 9 rows of `slots` x 256 floats, groups * slots <= 8 * 256 waves
constexpr int64_t kF34ScratchFloats = static_cast<int64_t>(9) * 8 * kF34WavesPerXcd * 256;

inline bool wgrad_f34_ok(const WgradParams& p, long min_tiles = kF34MinTiles) {
    const bool aligned = (p.w % 4 == 0) && (p.h % 16 == 0) && (p.dy_w % 4 == 0) && (p.dy_cs % 4 == 0) && (p.dy_ns % 4 == 0) && (p.in_w % 4 == 0) &&
                         (p.in_cs % 4 == 0) && (p.in_ns % 4 == 0) && (reinterpret_cast<uintptr_t>(p.dy) % 16 == 0) &&
                         (reinterpret_cast<uintptr_t>(p.in) % 16 == 0);
    const long quads = static_cast<long>((p.w + 15) / 16) * (p.h / 16) * p.n;
    // (the buffer descriptors address one sample with 32-bit byte offsets)
    const bool small = static_cast<int64_t>(p.in_cs) * (p.cin + 16) * 4 < (1ll << 31) && static_cast<int64_t>(p.dy_cs) * 12 * 4 < (1ll << 31);
    return aligned && small && p.cout == 12 && p.cin >= 16 && (p.cin + 15) / 16 <= kF34WavesPerXcd && quads * 16 >= min_tiles;
}
// the first convolution: a raw input of at most 16 channels, output channels in sets of 12
inline bool wgrad_f34_raw_ok(const WgradParams& p, long min_tiles = kF34MinTiles) {
    WgradParams q = p;
    q.cout = 12; q.cin = 16;
    return p.cin <= 16 && p.cout % 12 == 0 && p.cout / 12 <= 16 && static_cast<int64_t>(p.in_cs) * 32 * 4 < (1ll << 31) && wgrad_f34_ok(q, min_tiles);
}

// waves per group and segment length: the longest segments (fewest restarts) among those that keep the waves evenly loaded
inline F34Plan wgrad_f34_plan(const WgradParams& p, int waves_per_xcd = kF34WavesPerXcd, bool raw = false) {
    F34Plan plan{};
    plan.groups = raw ? p.cout / 12 : (p.cin + 15) / 16;
    const int CY = p.h / 16, S = (p.w + 15) / 16;
    float best = 1e30f;
    for (int segq = 4; segq >= 1; segq >>= 1) {
        if (CY % segq) continue;
        const int per_xcd = (p.n * (CY / segq) * S + 7) / 8;
        const int wpg = std::min(waves_per_xcd / plan.groups, per_xcd);
        const int iters = (per_xcd + wpg - 1) / wpg;
        // time ~ iterations x (steps of a segment + its start-up)
        const float cost = iters * (4.f * segq + 1.5f);
        if (cost < best * 0.97f) { best = cost; plan.segq = segq; plan.wpg = wpg; }
    }
    plan.spb = 1;
    for (int g = 0; g < plan.groups; ++g) plan.spb = std::max(plan.spb, f34_block_count(plan, g));
    plan.slots = 8 * plan.spb;
    return plan;
}

// blocks: 512 = two per CU (the default), 256 = one per CU (in-job A/B: leaves half of every CU's registers to the other stream's kernels)
// batch: the launch leaves its partial sums in `scratch` and records what their reduction needs instead of reducing them
// (launch_wgrad_f34_reduce_batch later, on the same stream; every launch of a batch needs its own scratch slice)
template <int EXP = 0, bool RAW = false, bool PREP = false>
inline int launch_wgrad_f34(const WgradParams& p, float* scratch, hipStream_t stream, int blocks = kF34Blocks, F34ReduceBatch* batch = nullptr) {
    static_assert(!PREP || RAW, "the fused gradient preparation belongs to the first convolution's form");
    const F34Plan plan = wgrad_f34_plan(p, blocks / 2, RAW);
    constexpr int lds = 4 * 2 * kF34Xs * 4;          // 67,584 bytes per block, two blocks per CU
    static bool configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!configured_by_device[dev & 15]) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_f34_kernel<EXP, RAW, PREP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured_by_device[dev & 15] = true;
    }
    wgrad_f34_kernel<EXP, RAW, PREP><<<blocks, kConvThreads, lds, stream>>>(p, scratch, plan);
    ENDO_LAUNCH_CHECK();
    if (batch && !RAW && batch->count < 4) {
        const int k = batch->count++;
        batch->partial[k] = scratch; batch->dw[k] = p.dw; batch->plan[k] = plan; batch->cin[k] = p.cin;
        return 0;
    }
    wgrad_f34_reduce_kernel<<<dim3(plan.groups * 9, 2), 256, 0, stream>>>(scratch, plan, p.cin, p.dw, RAW ? 1 : 0);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// (round 6: ONE slice -- a block adds all 8 * spb rows of its (group, tap) and issues one atomic per weight-gradient element; with 8 slices the
// eightfold atomics on dW cost more than the shorter loops saved: 503.7 -> 507.1 frame-pairs/s with 2 slices, 506 -> 508 with 1, profiles/r06_ab_runs.txt)
inline int launch_wgrad_f34_reduce_batch(F34ReduceBatch& batch, hipStream_t stream) {
    if (batch.count == 0) return 0;
    int gmax = 0;
    for (int k = 0; k < batch.count; ++k) gmax = std::max(gmax, batch.plan[k].groups);
    wgrad_f34_reduce_batch_kernel<<<dim3(gmax * 9, 1, batch.count), 256, 0, stream>>>(batch);
    ENDO_LAUNCH_CHECK();
    batch.count = 0;
    return 0;
}

}  // namespace endo
