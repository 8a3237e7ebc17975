// Data gradient of a growth-12 dense layer, fused with ReLU/BatchNorm backward -- persistent form.
//
// For this layer K is tiny (12 output maps x 9 taps = 108) and N is the layer's whole input width
// (48..372 channels): per pixel the kernel moves 3 * Cin * 4 bytes (read x, read-modify-write the
// gradient buffer) for 216 * Cin flops, i.e. 18 flop/byte -- it is HBM-bound.  So it
// is organised as a streaming kernel with a small matmul in the middle:
//   * one block per 32 x 8 pixel tile; the dY tile (12 maps + halo) is DMA'd to LDS ONCE
//   * the block then walks the input channels 16 at a time; per group: the 9x12x16 weight slice arrives
//     by LDS-DMA into a double buffer, the x / gradient-buffer operands of the NEXT group are loaded
//     into registers before the MFMAs of the current group (so ~8 KB per wave is always in flight),
//     108 MFMAs per wave, then the BN/ReLU epilogue of conv_kernels.h on registers
//   * per-channel sum dz / sum dz*xhat: wave shuffle -> LDS -> one fp64 atomic per channel per block
// Needs W % 4 == 0 (16-byte DMA / float4 epilogue); other shapes use the generic kernel.
#pragma once

#include "conv_dma_kernels.h"

namespace endo {

template <int WX, int R>
struct DgradGeom {
    using G = ConvGeom<3, 12, WX, R, 4>;
    static constexpr int kWG = 9 * 12 * 16;                                   // weights per 16-channel group
    static constexpr int kWPre = (kWG + kConvThreads - 1) / kConvThreads;     // 7
    static size_t bytes(int cout) {
        const int cap = (cout + 15) / 16 * 16;
        return sizeof(float) * (12 * G::kCS + 2 * kWG + 4 * cap + 2 * 4 * 16 * 2);
    }
};

template <int WX, int R>
__global__ void __launch_bounds__(kConvThreads) dgrad_dense_kernel(const ConvParams p0) {
    int grp, n;
    group_of(p0, blockIdx.z, grp, n);
    const ConvParams p = group_view(p0, grp);
    using D = DgradGeom<WX, R>;
    using G = typename D::G;
    static_assert(G::kPos == 1, "one 16-byte unit per thread per channel");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_dy = smem;                               // [12][kCS]
    float* s_w = s_dy + 12 * G::kCS;                  // [2][9][12][16]
    float* s_cst = s_w + 2 * D::kWG;                  // [cap][4] scale, beta, mean, rstd
    const int cap = (p.cout + 15) / 16 * 16;
    float* s_red = s_cst + 4 * cap;                   // [2][4 waves][16][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int tile = blockIdx.x;
    const int x0 = (tile % p.tiles_x) * G::kTileX;
    const int y0 = (tile / p.tiles_x) * G::kTileY;
    const int wx = (wave % WX) * 16;
    const int wy = (wave / WX) * R;
    const int px = x0 + wx + 4 * lk;
    const int ngroups = (p.cout + 15) / 16;

    // ---- per-channel BN constants for every input channel of the layer ----
    for (int c = tid; c < cap; c += kConvThreads) {
        float mean = 0.f, rstd = 0.f, scale = 0.f, beta = 0.f;
        if (c < p.cout) {
            mean = p.bn_saved[2 * c];
            rstd = p.bn_saved[2 * c + 1];
            scale = p.bn_gamma[c] * rstd;
            beta = p.bn_beta[c];
        }
        s_cst[4 * c] = scale; s_cst[4 * c + 1] = beta; s_cst[4 * c + 2] = mean; s_cst[4 * c + 3] = rstd;
    }

    // ---- dY tile: 12 maps, one float4 per thread per map ----
    {
        const int e = tid;
        const float* src_pad = g_pad_consts + 4;
        int goff = 0;
        bool ok = false;
        if (e < G::kUnits) {
            const int ry = e / (G::kCols / 4);
            const int rx = (e - ry * (G::kCols / 4)) * 4;
            const int gy = y0 - 1 + ry;
            const int gx = x0 - G::kLeft + rx;
            ok = gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
            goff = gy * p.in_w + gx;
        }
        const float* in_n = p.in + n * p.in_ns;
        const int e0 = wave * 64;
        if (e0 < G::kUnits) {
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                const float* src = ok ? in_n + static_cast<int64_t>(c) * p.in_cs + goff : src_pad;
                if (e0 + lane < G::kUnits)
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_dy + c * G::kCS + 4 * e0), 16, 0, 0);
            }
        }
    }

    // ---- weight slice addressing: element (tap, c, j) of a group <- W[c][co_base + j][8 - tap] ----
    int woff[D::kWPre];
    int wj[D::kWPre];
#pragma unroll
    for (int k = 0; k < D::kWPre; ++k) {
        const int e = tid + k * kConvThreads;
        const int j = e % 16;
        const int rest = e / 16;
        const int c = rest % 12;
        const int tap = rest / 12;
        woff[k] = (c * p.w_cin + j) * 9 + (8 - tap);
        wj[k] = j;
    }
    auto issue_weights = [&](int g, int buf) {
        const int co_base = g * 16;
#pragma unroll
        for (int k = 0; k < D::kWPre; ++k) {
            const int e0 = k * kConvThreads + wave * 64;
            if (e0 < D::kWG) {
                const bool ok = (e0 + lane < D::kWG) && (co_base + wj[k] < p.cout);
                const float* src = ok ? p.wgt + woff[k] + co_base * 9 : g_pad_consts + 4;
                if (e0 + lane < D::kWG)
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_w + buf * D::kWG + e0), 4, 0, 0);
            }
        }
    };

    // ---- epilogue operands of a group: x and the gradient-buffer values it accumulates into ----
    const float* x_n = p.x + n * p.x_ns;
    float* out_n = p.out + n * p.out_ns;
    auto load_operands = [&](int g, f32x4 (&xv)[R], f32x4 (&dv)[R]) {
        const int co = g * 16 + li;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int y = y0 + wy + r;
            xv[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            dv[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (co < p.cout && y < p.h && px + 3 < p.w) {
                xv[r] = *reinterpret_cast<const f32x4*>(x_n + static_cast<int64_t>(co) * p.x_cs + y * p.out_w + px);
                if (co >= p.acc_from) dv[r] = *reinterpret_cast<const f32x4*>(out_n + static_cast<int64_t>(co) * p.out_cs + y * p.out_w + px);
            }
        }
    };

    f32x4 xc[R], dc[R], xn[R], dn[R];
    issue_weights(0, 0);
    load_operands(0, xc, dc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int g = 0; g < ngroups; ++g) {
        const int buf = g & 1;
        if (g + 1 < ngroups) {
            load_operands(g + 1, xn, dn);
            issue_weights(g + 1, buf ^ 1);
        }
        // ---- 108 MFMAs: K = 3 channel quads x 9 taps ----
        f32x4 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* wb = s_w + buf * D::kWG;
#pragma unroll
        for (int quad = 0; quad < 3; ++quad) {
            const float* a_base = s_dy + (quad * 4 + lk) * G::kCS + wy * G::kCols + wx + li + G::kColOff;
            const float* b_base = wb + (quad * 4 + lk) * 16 + li;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float a[R + 2];
#pragma unroll
                for (int r = 0; r < R + 2; ++r) a[r] = a_base[r * G::kCols + dx];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float b = b_base[(dy * 3 + dx) * 12 * 16];
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r + dy], b, acc[r], 0, 0, 0);
                }
            }
        }
        // ---- ReLU mask + BN backward on registers (see conv_kernels.h EPI_DGRAD_BN) ----
        {
            const int co = g * 16 + li;
            const float scale = s_cst[4 * co], beta = s_cst[4 * co + 1], mean = s_cst[4 * co + 2], rstd = s_cst[4 * co + 3];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = y0 + wy + r;
                if (co < p.cout && y < p.h && px + 3 < p.w) {
                    f32x4 o = dc[r];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xcen = xc[r][e] - mean;
                        const float z = fmaf(xcen, scale, beta);
                        const float dz = z > 0.f ? acc[r][e] : 0.f;
                        s1 += dz;
                        s2 += dz * (xcen * rstd);
                        o[e] += scale * dz;
                    }
                    *reinterpret_cast<f32x4*>(out_n + static_cast<int64_t>(co) * p.out_cs + y * p.out_w + px) = o;
                }
            }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lk == 0) {
                float* red = s_red + buf * (4 * 16 * 2);
                red[(wave * 16 + li) * 2] = s1;
                red[(wave * 16 + li) * 2 + 1] = s2;
            }
        }
        if (g + 1 < ngroups) {
#pragma unroll
            for (int r = 0; r < R; ++r) { xc[r] = xn[r]; dc[r] = dn[r]; }
        }
        // next group's weights landed (issued a whole MFMA phase ago); publish them and the reductions
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < 32) {
            const int j = tid >> 1, which = tid & 1;
            const int co = g * 16 + j;
            if (co < p.cout) {
                const float* red = s_red + buf * (4 * 16 * 2);
                double t = 0.0;
                for (int wv = 0; wv < 4; ++wv) t += static_cast<double>(red[(wv * 16 + j) * 2 + which]);
                atomicAdd(p.bn_scratch + bn_slot_offset(p.bn_slot_stride) + 2 * co + which, t);
            }
        }
    }
}

template <int WX, int R>
inline int launch_dgrad_dense(ConvParams p, hipStream_t stream) {
    using D = DgradGeom<WX, R>;
    using G = typename D::G;
    p.tiles_x = (p.w + G::kTileX - 1) / G::kTileX;
    const int tiles_y = (p.h + G::kTileY - 1) / G::kTileY;
    const size_t smem = D::bytes(p.cout);
    static size_t configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& configured = configured_by_device[dev & 15];
    if (smem > 48 * 1024 && smem > configured) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_dense_kernel<WX, R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(smem)));
        configured = smem;
    }
    dgrad_dense_kernel<WX, R><<<dim3(p.tiles_x * tiles_y, 1, p.n), kConvThreads, smem, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// persistent kernel where its preconditions hold and there are enough tiles to fill the chip,
// the generic LDS-DMA kernel otherwise
inline int launch_dgrad_dense_auto(const ConvParams& p, hipStream_t stream) {
    const bool aligned = (p.w % 4 == 0) && (p.in_w % 4 == 0) && (p.in_cs % 4 == 0) && (p.in_ns % 4 == 0) && (p.out_w % 4 == 0) &&
                         (p.out_cs % 4 == 0) && (p.out_ns % 4 == 0) && (p.x_cs % 4 == 0) && (p.x_ns % 4 == 0) && p.cin == 12 &&
                         (reinterpret_cast<uintptr_t>(p.in) % 16 == 0) && (reinterpret_cast<uintptr_t>(p.out) % 16 == 0) &&
                         (reinterpret_cast<uintptr_t>(p.x) % 16 == 0);
    if (aligned) {
        const long tiles = static_cast<long>((p.w + 31) / 32) * ((p.h + 7) / 8) * p.n;
        if (tiles >= 512) return launch_dgrad_dense<2, 4>(p, stream);
    }
    return launch_conv_dma_auto<3, 12, 1, IN_PLAIN, EPI_DGRAD_BN, 4, 1, 1>(p, stream);
}

}  // namespace endo
