// Weight gradient of the transition-up convolution (nearest x2 -> conv3x3 48 -> 48, reference models.py:70-80) in
// sub-pixel form.  The forward pass of output phase (a, b) is a 2x2 convolution of the LOW-resolution input with
// tap-summed weights W_eff (conv_dma_kernels.h, PH), so
//     dW_eff[(a,b,co)][ci][t] = sum over low-res pixels p of dY[co][2 p + (a,b)] * x[ci][p + off_(a,b)(t)]       (16 of them)
//     dW[co][ci][ky][kx]      = sum over the four phases of dW_eff[(a,b,co)][ci][t containing (ky,kx)]
// -- 16 instead of 36 pixel-sized correlations (4/9 of the MACs), and the x2 gather disappears.
//
// Kernel shape = wgrad_nsplit_kernels.h with the roles the data dictates: the SHIFTED operand is the low-resolution
// activation (a 3-row x 40-column window of the block's 24 input channels, LDS-DMA, two buffers), M = (ci, t) = 96 rows
// = 6 MFMA row groups; the other operand, dY, is used unshifted and by one lane only, so each lane loads its 8
// consecutive full-resolution values per (channel group, pixel quad) straight into registers -- both column phases
// come out of the same two float4s.  Wave w of a block owns phase (a, b) = (w >> 1, w & 1) and all 48 couts (3 groups):
// 18 accumulators, 6 LDS reads per 18 MFMAs.  blockIdx.y = half of the input channels.  Blocks own contiguous chunk
// ranges (chunk = one low-res row segment of 32 pixels) and write partial sums; tu_wgrad_subpix_reduce_kernel adds
// them and scatters every dW_eff entry onto the 1, 2 or 4 original taps it stands for.
#pragma once

#include "conv_dma_kernels.h"
#include "wgrad_kernels.h"

namespace endo {

constexpr int kSpSeg = 32;                         // low-res pixels per chunk
constexpr int kSpCols = kSpSeg + 8;                // window columns: 4-pixel aligned halo on both sides
constexpr int kSpMap = 3 * kSpCols;                // floats per channel window (rows y-1, y, y+1)
constexpr int kSpCh = 24;                          // input channels per block
constexpr int kSpBuf = kSpCh * kSpMap;             // floats per buffer
constexpr int kSpMG = kSpCh * 4 / 16;              // 6 MFMA row groups: m = ci_local * 4 + t
constexpr int kSpUnits = kSpBuf / 4;               // 720 float4 per window

__global__ void __launch_bounds__(kConvThreads) tu_wgrad_subpix_kernel(const WgradParams p, float* __restrict__ partial, int chunks_per_block) {
    __shared__ __attribute__((aligned(16))) float smem[2 * kSpBuf];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int pa = wave >> 1, pb = wave & 1;           // this wave's output phase
    const int ci_base = blockIdx.y * kSpCh;
    const int segs = (p.w + kSpSeg - 1) / kSpSeg;      // p.h, p.w: the LOW-resolution grid
    const int chunks_total = segs * p.h * p.n;
    const int c_begin = blockIdx.x * chunks_per_block;
    const int c_end = min(c_begin + chunks_per_block, chunks_total);

    // row m = 16 g + li = ci_local * 4 + t reads x[ci][y + oy][x + ox]: window row oy + 1, window column x - x0 + 4 + ox,
    // with (oy, ox) the t-th of the 2 x 2 offsets of phase (a, b): a = 0 -> rows (y-1, y), a = 1 -> (y, y+1); columns likewise
    int aoff[kSpMG];
#pragma unroll
    for (int g = 0; g < kSpMG; ++g) {
        const int m = 16 * g + li;
        const int cl = m >> 2, t = m & 3;
        const int oy = (t >> 1) + pa - 1, ox = (t & 1) + pb - 1;
        aoff[g] = cl * kSpMap + (oy + 1) * kSpCols + 4 + ox + 4 * lk;
    }

    f32x4 acc[3][kSpMG];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int m = 0; m < kSpMG; ++m) acc[g][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* pad_zero = g_pad_consts + 4;
    // this thread's units of the activation window (chunk independent)
    constexpr int kSlots = (kSpUnits + kConvThreads - 1) / kConvThreads;      // 3
    int w_off[kSlots], w_row[kSlots], w_col[kSlots];
#pragma unroll
    for (int k = 0; k < kSlots; ++k) {
        const int e = k * kConvThreads + tid;
        const int ch = e / (kSpMap / 4);
        const int r = e - ch * (kSpMap / 4);
        w_row[k] = r / (kSpCols / 4);
        w_col[k] = 4 * (r - w_row[k] * (kSpCols / 4));
        w_off[k] = (ci_base + ch) * p.in_cs + w_row[k] * p.in_w + w_col[k];
    }
    int64_t d_off[3];          // this lane's cout rows of dY
#pragma unroll
    for (int g = 0; g < 3; ++g) d_off[g] = static_cast<int64_t>(16 * g + li) * p.dy_cs + 8 * lk;

    int i_n = c_begin / (segs * p.h);
    int i_y = (c_begin - i_n * segs * p.h) / segs;
    int i_seg = c_begin - (i_n * p.h + i_y) * segs;

    f32x4 raw[3][2][2];        // dY of the chunk in flight: [cout group][pixel quad][low / high float4 of the 8 values]
    unsigned raw_ok = 0;

    auto issue = [&](int buf) {
        const int x0 = i_seg * kSpSeg;
        const WgSample sm(p, i_n);
        float* s_x = smem + buf * kSpBuf;
        const float* x_base = p.in + sm.in_off(p) + static_cast<int64_t>(i_y - 1) * p.in_w + x0 - 4;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            const int e0 = k * kConvThreads + wave * 64;
            if (e0 < kSpUnits) {
                const bool ok = static_cast<unsigned>(i_y - 1 + w_row[k]) < static_cast<unsigned>(p.h) &&
                                static_cast<unsigned>(x0 - 4 + w_col[k]) < static_cast<unsigned>(p.w);
                const float* src = ok ? x_base + w_off[k] : pad_zero;
                if (e0 + lane < kSpUnits) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_x + 4 * e0), 16, 0, 0);
            }
        }
        // full-resolution row 2 y + a, columns 2 (x0 + 16 q + 4 lk) .. + 7
        const float* d_base = p.dy + sm.dy_off(p) + static_cast<int64_t>(2 * i_y + pa) * p.dy_w + 2 * x0;
        raw_ok = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (x0 + 16 * q + 4 * lk < p.w) raw_ok |= 1u << q;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float* src = (raw_ok & (1u << q)) ? d_base + d_off[g] + 32 * q : pad_zero;
                raw[g][q][0] = *reinterpret_cast<const f32x4*>(src);
                raw[g][q][1] = *reinterpret_cast<const f32x4*>((raw_ok & (1u << q)) ? src + 4 : pad_zero);
            }
        if (++i_seg == segs) {
            i_seg = 0;
            if (++i_y == p.h) { i_y = 0; ++i_n; }
        }
    };

    if (c_begin < c_end) issue(0);
    int buf = 0;
    for (int chunk = c_begin; chunk < c_end; ++chunk, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // this wave's column phase of the 8 loaded values: elements b, b + 2, b + 4, b + 6
        f32x4 bv[3][2];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool ok = raw_ok & (1u << q);
                bv[g][q][0] = ok ? (pb ? raw[g][q][0][1] : raw[g][q][0][0]) : 0.f;
                bv[g][q][1] = ok ? (pb ? raw[g][q][0][3] : raw[g][q][0][2]) : 0.f;
                bv[g][q][2] = ok ? (pb ? raw[g][q][1][1] : raw[g][q][1][0]) : 0.f;
                bv[g][q][3] = ok ? (pb ? raw[g][q][1][3] : raw[g][q][1][2]) : 0.f;
            }
        if (chunk + 1 < c_end) issue(buf ^ 1);

        const float* s_x = smem + buf * kSpBuf;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a[kSpMG];
#pragma unroll
                for (int m = 0; m < kSpMG; ++m) a[m] = s_x[aoff[m] + 16 * q + e];
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int m = 0; m < kSpMG; ++m)
                        acc[g][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bv[g][q][e], acc[g][m], 0, 0, 0);
            }
    }

    // partial[((((bx * 2 + by) * 4 + wave) * 3 + g) * 6 + m) * 4 + r][lane] = D[row 16 m + 4 lk + r][cout 16 g + li]
    float* out = partial + ((static_cast<int64_t>(blockIdx.x) * gridDim.y + blockIdx.y) * 4 + wave) * (3 * kSpMG * 256);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int m = 0; m < kSpMG; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[((g * kSpMG + m) * 4 + r) * 64 + lane] = acc[g][m][r];
}

// grid (2 * 4 * 3 * 6 = 144 row blocks, slices of the partial blocks): sum, then scatter onto the original taps
__global__ void __launch_bounds__(256) tu_wgrad_subpix_reduce_kernel(const float* __restrict__ partial, int blocks, int cin, float* __restrict__ dw) {
    const int rb = blockIdx.x;                         // ((by * 4 + wave) * 3 + g) * 6 + m
    const int m6 = rb % kSpMG, g = (rb / kSpMG) % 3, wave = (rb / (kSpMG * 3)) % 4, by = rb / (kSpMG * 3 * 4);
    const int per = (blocks + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(blocks, b0 + per);
    const int64_t stride = static_cast<int64_t>(2) * 4 * 3 * kSpMG * 256;
    const float* src = partial + (static_cast<int64_t>(by * 4 + wave) * 3 * kSpMG + g * kSpMG + m6) * 256 + threadIdx.x;
    float s = 0.f;
    for (int b = b0; b < b1; ++b) s += src[b * stride];
    if (b0 >= b1) return;
    const int lane = threadIdx.x & 63, r = threadIdx.x >> 6;
    const int m = 16 * m6 + 4 * (lane >> 4) + r;
    const int ci = by * kSpCh + (m >> 2), t = m & 3;
    const int co = 16 * g + (lane & 15);
    const int pa = wave >> 1, pb = wave & 1;
    // original taps summed into effective tap (tyi, txi) of phase (a, b)  (tu_phase_weights_kernel)
    const int tyi = t >> 1, txi = t & 1;
    const int ky0 = pa == 0 ? (tyi == 0 ? 0 : 1) : (tyi == 0 ? 0 : 2), ky1 = pa == 0 ? (tyi == 0 ? 0 : 2) : (tyi == 0 ? 1 : 2);
    const int kx0 = pb == 0 ? (txi == 0 ? 0 : 1) : (txi == 0 ? 0 : 2), kx1 = pb == 0 ? (txi == 0 ? 0 : 2) : (txi == 0 ? 1 : 2);
    float* dst = dw + (static_cast<int64_t>(co) * cin + ci) * 9;
    for (int ky = ky0; ky <= ky1; ++ky)
        for (int kx = kx0; kx <= kx1; ++kx) atomicAdd(dst + ky * 3 + kx, s);
}

constexpr int kSpMaxBlocks = 384;
constexpr int64_t kSpScratchFloats = static_cast<int64_t>(kSpMaxBlocks) * 2 * 4 * 3 * kSpMG * 256;      // 14.2 M floats

// p: h, w = the LOW-resolution grid; in = low-res activations (48 channels); dy = full-resolution gradient (48 maps)
inline bool tu_wgrad_subpix_ok(const WgradParams& p) {
    return p.cin == 2 * kSpCh && p.cout == 48 && (p.w % 4 == 0) && (p.in_w % 4 == 0) && (p.in_cs % 4 == 0) && (p.in_ns % 4 == 0) &&
           (p.dy_w % 8 == 0) && (p.dy_cs % 4 == 0) && (p.dy_ns % 4 == 0) && (reinterpret_cast<uintptr_t>(p.in) % 16 == 0) &&
           (reinterpret_cast<uintptr_t>(p.dy) % 16 == 0);
}

inline int launch_tu_wgrad_subpix(const WgradParams& p, float* scratch, hipStream_t stream) {
    const int chunks_total = ((p.w + kSpSeg - 1) / kSpSeg) * p.h * p.n;
    int blocks = kSpMaxBlocks;
    if (blocks > chunks_total) blocks = chunks_total;
    const int per = (chunks_total + blocks - 1) / blocks;
    blocks = (chunks_total + per - 1) / per;
    tu_wgrad_subpix_kernel<<<dim3(blocks, 2), kConvThreads, 0, stream>>>(p, scratch, per);
    ENDO_LAUNCH_CHECK();
    tu_wgrad_subpix_reduce_kernel<<<dim3(2 * 4 * 3 * kSpMG, 2), 256, 0, stream>>>(scratch, blocks, p.cin, p.dw);
    ENDO_LAUNCH_CHECK();
    return 0;
}

}  // namespace endo
