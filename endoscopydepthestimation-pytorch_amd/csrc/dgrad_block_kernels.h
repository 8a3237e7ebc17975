// Data gradient of a whole dense block (up to 4 growth-12 layers) for a range of the block's input
// channels, fused with the ReLU / BatchNorm backward of every layer -- the HBM-traffic fix for dgrad.
//
// Layer l of a dense block reads base ++ new_0 .. new_{l-1}; every layer therefore sends a gradient
// into the SAME base channels.  Layer by layer that is 4 passes of (read x, read-modify-write the
// gradient buffer) over the base -- and the kernel is HBM-bound (18 flop/byte).  But the sequential
// dependency between the layers only runs through the 36 NEW channels: once G_3..G_0 (the finalised
// output gradients of the 4 layers, 48 maps) are known, the base channels can be done in ONE pass:
//     dbuf[c] (+)= sum_l scale_l[c] * [z_l[c] > 0] * convT_l(G_l)[c]
// with x[c] read once, dbuf[c] read and written once, and the per-layer BN sums (sum dz, sum dz*xhat)
// reduced on the fly.  On a 144-channel base this is 3x less HBM traffic than 4 separate dgrads.
// The same kernel with NL = 1 and a 12/24/36-channel range serves the "new channel" dgrads that
// carry the layer-to-layer dependency.
//
// Structure (as dgrad_kernels.h): block = one 32 x 6 pixel tile; the NL*12 G maps (+halo) are DMA'd to
// LDS once; then steps s = (16-channel group, layer): 9x12x16 weight slice by LDS-DMA into a double
// buffer, 27 * 3 MFMAs per wave, masked accumulate into `total`; after the last layer of a group the
// float4 epilogue stores.  x / dbuf of the next group and the BN constants of the next step are loaded
// one step ahead.  One barrier per step.
#pragma once

#include "conv_dma_kernels.h"

namespace endo {

constexpr int kMaxFusedLayers = 4;

struct DgradBlockParams {
    int n, h, w, tiles_x;
    // G: NL * 12 consecutive maps of the gradient buffer (finalised), layer l at maps [12 l, 12 l + 12)
    const float* g;
    int64_t g_ns;
    int g_cs, g_w;
    // the channel range this launch produces
    const float* x;          // activations of the range
    float* out;              // gradient buffer of the range
    int64_t ns;
    int cs;
    int count;               // channels in the range
    int acc_from;            // channels >= acc_from accumulate into out, others overwrite
    int w_ci_off;            // index of the range's first channel inside each layer's input
    // per layer (pointers already offset to the range's first channel where per-channel)
    const float* wgt[kMaxFusedLayers];
    int w_cin[kMaxFusedLayers];
    const float* saved[kMaxFusedLayers];
    const float* gamma[kMaxFusedLayers];
    const float* beta[kMaxFusedLayers];
    double* scratch[kMaxFusedLayers];
};

template <int NL, int WX, int R>
struct DgradBlockGeom {
    static constexpr int kTileX = 16 * WX;
    static constexpr int kTileY = R * (4 / WX);
    static constexpr int kRows = kTileY + 2;
    static constexpr int kCols = kTileX + 2;
    static constexpr int kPlane = kRows * kCols;
    static constexpr int kCS = ((kPlane - 16 + 31) / 32) * 32 + 16;
    static constexpr int kPos = (kPlane + kConvThreads - 1) / kConvThreads;
    static constexpr int kWG = 9 * 12 * 16;
    static constexpr int kWPre = (kWG + kConvThreads - 1) / kConvThreads;
    static constexpr size_t kBytes = sizeof(float) * (NL * 12 * kCS + 2 * kWG + 2 * 4 * 16 * 2);
    static_assert(kBytes <= 80 * 1024, "two blocks per CU");
};

template <int NL, int WX, int R>
__global__ void __launch_bounds__(kConvThreads) dgrad_block_kernel(const DgradBlockParams p) {
    using G = DgradBlockGeom<NL, WX, R>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_g = smem;                               // [NL*12][kCS]
    float* s_w = s_g + NL * 12 * G::kCS;             // [2][9][12][16]
    float* s_red = s_w + 2 * G::kWG;                 // [2][4 waves][16][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int tile = (gridDim.x & 7) == 0 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int x0 = (tile % p.tiles_x) * G::kTileX;
    const int y0 = (tile / p.tiles_x) * G::kTileY;
    const int n = blockIdx.z;
    const int wx = (wave % WX) * 16;
    const int wy = (wave / WX) * R;
    const int px = x0 + wx + 4 * lk;
    const int ngroups = (p.count + 15) / 16;
    const int nsteps = ngroups * NL;

    // ---- G tiles: NL*12 maps with a 1-pixel halo, dword DMA (once per block) ----
    {
        int goff[G::kPos];
        unsigned ok = 0;
#pragma unroll
        for (int k = 0; k < G::kPos; ++k) {
            const int e = tid + k * kConvThreads;
            goff[k] = 0;
            if (e < G::kPlane) {
                const int ry = e / G::kCols, rx = e - ry * G::kCols;
                const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
                if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) { ok |= 1u << k; goff[k] = gy * p.g_w + gx; }
            }
        }
        const float* g_n = p.g + n * p.g_ns;
        for (int c = 0; c < NL * 12; ++c) {
            const float* plane = g_n + static_cast<int64_t>(c) * p.g_cs;
#pragma unroll
            for (int k = 0; k < G::kPos; ++k) {
                const int e0 = k * kConvThreads + wave * 64;
                if (e0 < G::kPlane) {
                    const float* src = (ok & (1u << k)) ? plane + goff[k] : g_pad_consts + 4;
                    if (e0 + lane < G::kPlane)
                        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_g + c * G::kCS + e0), 4, 0, 0);
                }
            }
        }
    }

    // ---- weight slice of step (group, layer): element (tap, c, j) <- W_l[c][w_ci_off + 16 group + j][8 - tap] ----
    int wc[G::kWPre], wrest[G::kWPre], wj[G::kWPre];
#pragma unroll
    for (int k = 0; k < G::kWPre; ++k) {
        const int e = tid + k * kConvThreads;
        const int j = e % 16;
        const int rest = e / 16;
        const int c = rest % 12;
        const int tap = rest / 12;
        wc[k] = c * 9;
        wrest[k] = j * 9 + (8 - tap);
        wj[k] = j;
    }
    auto issue_weights = [&](int step, int buf) {
        const int grp = step / NL, l = step - grp * NL;
        const int co_base = grp * 16;
        const float* wl = p.wgt[l] + static_cast<int64_t>(p.w_ci_off + co_base) * 9;
        const int wcin = p.w_cin[l];
#pragma unroll
        for (int k = 0; k < G::kWPre; ++k) {
            const int e0 = k * kConvThreads + wave * 64;
            if (e0 < G::kWG) {
                const bool ok = (e0 + lane < G::kWG) && (co_base + wj[k] < p.count);
                const float* src = ok ? wl + wc[k] * wcin + wrest[k] : g_pad_consts + 4;
                if (e0 + lane < G::kWG)
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_w + buf * G::kWG + e0), 4, 0, 0);
            }
        }
    };

    // ---- per-lane BN constants of a step: channel co = 16 group + li of layer l ----
    auto load_consts = [&](int step, float (&cst)[4]) {
        const int grp = step / NL, l = step - grp * NL;
        const int co = grp * 16 + li;
        cst[0] = cst[1] = cst[2] = cst[3] = 0.f;
        if (co < p.count) {
            const float mean = p.saved[l][2 * co], rstd = p.saved[l][2 * co + 1];
            cst[0] = p.gamma[l][co] * rstd;     // scale
            cst[1] = p.beta[l][co];
            cst[2] = mean;
            cst[3] = rstd;
        }
    };

    // ---- epilogue operands of a group ----
    const float* x_n = p.x + n * p.ns;
    float* out_n = p.out + n * p.ns;
    auto load_operands = [&](int grp, f32x4 (&xv)[R], f32x4 (&dv)[R]) {
        const int co = grp * 16 + li;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int y = y0 + wy + r;
            xv[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            dv[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (co < p.count && y < p.h && px + 3 < p.w) {
                xv[r] = *reinterpret_cast<const f32x4*>(x_n + static_cast<int64_t>(co) * p.cs + y * p.w + px);
                if (co >= p.acc_from) dv[r] = *reinterpret_cast<const f32x4*>(out_n + static_cast<int64_t>(co) * p.cs + y * p.w + px);
            }
        }
    };

    f32x4 xc[R], dc[R], xn[R], dn[R], total[R];
    float cc[4], cn[4];
#pragma unroll
    for (int r = 0; r < R; ++r) total[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue_weights(0, 0);
    load_consts(0, cc);
    load_operands(0, xc, dc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const int grp = step / NL, l = step - grp * NL;
        const bool last_layer = (l == NL - 1);
        if (step + 1 < nsteps) {
            if (last_layer) load_operands(grp + 1, xn, dn);
            load_consts(step + 1, cn);
            issue_weights(step + 1, buf ^ 1);
        }
        // ---- convT_l(G_l) for 16 channels: K = 3 map quads x 9 taps ----
        f32x4 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* wb = s_w + buf * G::kWG;
#pragma unroll
        for (int quad = 0; quad < 3; ++quad) {
            const float* a_base = s_g + (l * 12 + quad * 4 + lk) * G::kCS + wy * G::kCols + wx + li;
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
        // ---- layer l's ReLU mask + BN backward, accumulated over the layers of the block ----
        {
            const int co = grp * 16 + li;
            const float scale = cc[0], beta = cc[1], mean = cc[2], rstd = cc[3];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = y0 + wy + r;
                if (co < p.count && y < p.h && px + 3 < p.w) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xcen = xc[r][e] - mean;
                        const float z = fmaf(xcen, scale, beta);
                        const float dz = z > 0.f ? acc[r][e] : 0.f;
                        s1 += dz;
                        s2 += dz * (xcen * rstd);
                        total[r][e] += scale * dz;
                    }
                    if (last_layer) {
                        f32x4 o = dc[r];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += total[r][e];
                        *reinterpret_cast<f32x4*>(out_n + static_cast<int64_t>(co) * p.cs + y * p.w + px) = o;
                    }
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
        if (last_layer) {
#pragma unroll
            for (int r = 0; r < R; ++r) { total[r] = f32x4{0.f, 0.f, 0.f, 0.f}; xc[r] = xn[r]; dc[r] = dn[r]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) cc[i] = cn[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < 32) {
            const int j = tid >> 1, which = tid & 1;
            const int co = grp * 16 + j;
            if (co < p.count) {
                const float* red = s_red + buf * (4 * 16 * 2);
                double t = 0.0;
                for (int wv = 0; wv < 4; ++wv) t += static_cast<double>(red[(wv * 16 + j) * 2 + which]);
                atomicAdd(p.scratch[l] + 2 * co + which, t);
            }
        }
    }
}

template <int NL, int WX, int R>
inline int launch_dgrad_block(DgradBlockParams p, hipStream_t stream) {
    using G = DgradBlockGeom<NL, WX, R>;
    p.tiles_x = (p.w + G::kTileX - 1) / G::kTileX;
    const int tiles_y = (p.h + G::kTileY - 1) / G::kTileY;
    static bool configured = false;
    if (!configured && G::kBytes > 48 * 1024) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_block_kernel<NL, WX, R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(G::kBytes)));
        configured = true;
    }
    dgrad_block_kernel<NL, WX, R><<<dim3(p.tiles_x * tiles_y, 1, p.n), kConvThreads, G::kBytes, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// float4 epilogue: W % 4 == 0 and 16-byte aligned planes
inline bool dgrad_block_ok(const DgradBlockParams& p) {
    return (p.w % 4 == 0) && (p.cs % 4 == 0) && (p.ns % 4 == 0) && (reinterpret_cast<uintptr_t>(p.x) % 16 == 0) &&
           (reinterpret_cast<uintptr_t>(p.out) % 16 == 0);
}

}  // namespace endo
