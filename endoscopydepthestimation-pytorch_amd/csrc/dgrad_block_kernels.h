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
// float4 epilogue results are parked in registers and stored at the start of the NEXT step (so that no store is
// in flight at the vmcnt(0) that publishes the next weight slice).  x / dbuf of a group set are loaded at its first
// layer's step, the BN constants one step ahead.  One barrier per step.
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
    int64_t slot_stride;      // copies of the sums, this many doubles apart (common.h: kBnSlots); 0 = one copy
    // grouped batch (see ConvParams): blockIdx.z = group * group_n + sample; g, x, out, saved, scratch move by group * gs floats
    int group_n;
    int64_t gs;
    // "virtual" old gradient (the last up block at the finest level): what `out` holds before this launch is the final 1x1
    // convolution's data gradient, g(pixel) * vw[channel] (reference models.py:167, 186: finalConv + abs) -- a rank-one tensor.  With vg set the
    // kernel forms it from the ONE plane g = grad_out * sign(pre) (vg: [group][sample][pixel], groups gs floats apart) and the 192
    // final-conv weights instead of reading 4 bytes per channel and pixel that another kernel would first have had to write
    // (192 planes = 1 GB at 16 x 256 x 320).  vw points at the range's first channel.
    const float* vg;
    const float* vw;
};

// VEC = 4: the G tile starts 4 pixels left of the output tile so that its rows are whole aligned float4s and arrive by 16-byte
// LDS-DMA (4x fewer DMA instructions: the tile load is most of a new-channel pass, which runs only NL steps on it).
template <int NL, int WX, int R, int GP, int VEC = 1>
struct DgradBlockGeom {
    static constexpr int kTileX = 16 * WX;
    static constexpr int kTileY = R * (4 / WX);
    static constexpr int kRows = kTileY + 2;
    static constexpr int kLeft = VEC == 4 ? 4 : 1;
    static constexpr int kCols = kTileX + 2 * kLeft;
    static constexpr int kColOff = kLeft - 1;          // fragment column = x - x0 + dx + kColOff
    static constexpr int kPlane = kRows * kCols;
    static constexpr int kUnits = kPlane / VEC;
    static constexpr int kCS = ((kPlane - 16 + 31) / 32) * 32 + 16;
    static constexpr int kPos = (kPlane + kConvThreads - 1) / kConvThreads;
    static constexpr int kNB = 16 * GP;                 // output channels per step
    static constexpr int kWG = 9 * 12 * kNB;
    static constexpr int kWPre = (kWG + kConvThreads - 1) / kConvThreads;
    static constexpr int kRed = GP * 4 * 16 * 2;        // per buffer: [GP][4 waves][16][2]
    static constexpr size_t kBytes = sizeof(float) * (NL * 12 * kCS + 2 * kWG + 2 * kRed);
    static_assert(kBytes <= 80 * 1024, "two blocks per CU");
};

// GP: 16-channel groups per step.  Two groups share every A (dY) fragment read and halve the barriers per MFMA, but
// measured (tools/conv_bench, level 0 / 1 / 2 shapes) GP = 2 is within +-3 % of GP = 1 and costs 80 more VGPRs;
// the library instantiates GP = 1.
// EXP: diagnostic bit mask for tools/conv_bench (0 in the library): 1 = no x / dbuf loads, 2 = no stores,
// 4 = weight slice loaded once, 8 = no BN-sum reduction, 16 = no dY tile load, 32 = epilogue reduced to an add
// BF: 1 = bf16 MFMA operands (ENDO_OPT_MFMA_BF16): the three row taps of a map and column tap in one v_mfma_f32_16x16x16_bf16
// (conv_dma_kernels.h, BF), fp32 accumulation and epilogue
template <int NL, int WX, int R, int GP, int EXP = 0, int PIPE = 1, int VEC = 1, int BF = 0>
__global__ void __launch_bounds__(kConvThreads) dgrad_block_kernel(const DgradBlockParams p0) {
    using G = DgradBlockGeom<NL, WX, R, GP, VEC>;
    const int grp = p0.group_n > 0 ? blockIdx.z / p0.group_n : 0;
    const int n = blockIdx.z - grp * p0.group_n;
    const DgradBlockParams& p = p0;
    const int64_t grp_off = grp * p0.gs;            // this group's tape / workspace offset (floats)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_g = smem;                               // [NL*12][kCS]
    float* s_w = s_g + NL * 12 * G::kCS;             // [2][9][12][16 GP]
    float* s_red = s_w + 2 * G::kWG;                 // [2][GP][4 waves][16][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int tile = (gridDim.x & 7) == 0 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int x0 = (tile % p.tiles_x) * G::kTileX;
    const int y0 = (tile / p.tiles_x) * G::kTileY;
    const int wx = (wave % WX) * 16;
    const int wy = (wave / WX) * R;
    const int px = x0 + wx + 4 * lk;
    // blockIdx.y: slice of the group sets (coarse levels have few tiles; the dY tile is re-read per slice, from L2)
    const int ngroups = (p.count + 15) / 16;
    const int gsets = (ngroups + GP - 1) / GP;
    const int gs_per = (gsets + gridDim.y - 1) / gridDim.y;
    const int gs_begin = blockIdx.y * gs_per;
    const int gs_end = min(gsets, gs_begin + gs_per);
    if (gs_begin >= gs_end) return;
    const int nsteps = (gs_end - gs_begin) * NL;

    // ---- G tiles: NL*12 maps with a 1-pixel halo (once per block) ----
    if constexpr (VEC == 4) {
        // 16-byte DMA: a map is kUnits float4 units; wave w takes maps w, w + 4, ...: one full instruction (64 units) and one for
        // the rest of the map, each writing contiguous LDS
        static_assert(G::kUnits > 64 && G::kUnits <= 128, "two DMA instructions per map");
        int off[2];
        bool ok[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = lane + 64 * k;
            const int ry = u / (G::kCols / 4), rx = (u - ry * (G::kCols / 4)) * 4;
            const int gy = y0 - 1 + ry, gx = x0 - G::kLeft + rx;
            ok[k] = u < G::kUnits && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;          // W % 4 == 0: a unit is all in or all out
            off[k] = ok[k] ? gy * p.g_w + gx : 0;
        }
        const float* g_n = p.g + grp_off + n * p.g_ns;
        for (int c = wave; c < ((EXP & 16) ? 0 : NL * 12); c += 4) {
            const float* plane = g_n + static_cast<int64_t>(c) * p.g_cs;
            __builtin_amdgcn_global_load_lds((gptr_t)(ok[0] ? plane + off[0] : g_pad_consts + 4), (lptr_t)(s_g + c * G::kCS), 16, 0, 0);
            if (lane + 64 < G::kUnits)
                __builtin_amdgcn_global_load_lds((gptr_t)(ok[1] ? plane + off[1] : g_pad_consts + 4), (lptr_t)(s_g + c * G::kCS + 256), 16, 0, 0);
        }
    } else {
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
        const float* g_n = p.g + grp_off + n * p.g_ns;
        for (int c = 0; c < ((EXP & 16) ? 0 : NL * 12); ++c) {
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

    // ---- weight slice of step (group set, layer): element (tap, c, j) <- W_l[c][w_ci_off + 16 GP gs + j][8 - tap] ----
    int wc[G::kWPre], wrest[G::kWPre], wj[G::kWPre];
#pragma unroll
    for (int k = 0; k < G::kWPre; ++k) {
        const int e = tid + k * kConvThreads;
        const int j = e % G::kNB;
        const int rest = e / G::kNB;
        const int c = rest % 12;
        const int tap = rest / 12;
        wc[k] = c * 9;
        wrest[k] = j * 9 + (8 - tap);
        wj[k] = j;
    }
    auto issue_weights = [&](int step, int buf) {
        const int gs = gs_begin + step / NL, l = step % NL;
        const int co_base = gs * G::kNB;
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

    // ---- per-lane BN constants of a step: channel co = 16 (GP gs + a) + li of layer l ----
    auto load_consts = [&](int step, float (&cst)[GP][4]) {
        const int gs = gs_begin + step / NL, l = step % NL;
#pragma unroll
        for (int a = 0; a < GP; ++a) {
            const int co = (gs * GP + a) * 16 + li;
            cst[a][0] = cst[a][1] = cst[a][2] = cst[a][3] = 0.f;
            if (co < p.count) {
                const float mean = p.saved[l][grp_off + 2 * co], rstd = p.saved[l][grp_off + 2 * co + 1];
                cst[a][0] = p.gamma[l][co] * rstd;     // scale
                cst[a][1] = p.beta[l][co];
                cst[a][2] = mean;
                cst[a][3] = rstd;
            }
        }
    };

    const float* x_n = p.x + grp_off + n * p.ns;
    float* out_n = p.out + grp_off + n * p.ns;
    f32x4 xc[GP][R], dc[GP][R], total[GP][R], po[GP][R];
    float cc[GP][4], cn[GP][4];
    int po_gs = -1;                                   // group set whose results wait in `po` (stored one step late)
    // epilogue operands of a group set; issued at its first layer's step and consumed at the end of that step
    auto load_operands = [&](int gs) {
#pragma unroll
        for (int a = 0; a < GP; ++a) {
            const int co = (gs * GP + a) * 16 + li;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = y0 + wy + r;
                xc[a][r] = f32x4{0.f, 0.f, 0.f, 0.f};
                dc[a][r] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!(EXP & 1) && co < p.count && y < p.h && px + 3 < p.w) {
                    xc[a][r] = *reinterpret_cast<const f32x4*>(x_n + static_cast<int64_t>(co) * p.cs + y * p.w + px);
                    if (p.vg) {          // (block-uniform) the old gradient is g * w_final[co]
                        const f32x4 gv = *reinterpret_cast<const f32x4*>(p.vg + grp_off + static_cast<int64_t>(n) * p.cs + y * p.w + px);
                        const float wf = p.vw[co];
                        dc[a][r] = f32x4{gv[0] * wf, gv[1] * wf, gv[2] * wf, gv[3] * wf};
                    } else if (co >= p.acc_from) dc[a][r] = *reinterpret_cast<const f32x4*>(out_n + static_cast<int64_t>(co) * p.cs + y * p.w + px);
                }
            }
        }
    };
    auto store_pending = [&]() {
#pragma unroll
        for (int a = 0; a < GP; ++a) {
            const int co = (po_gs * GP + a) * 16 + li;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = y0 + wy + r;
                if (!(EXP & 2) && co < p.count && y < p.h && px + 3 < p.w)
                    *reinterpret_cast<f32x4*>(out_n + static_cast<int64_t>(co) * p.cs + y * p.w + px) = po[a][r];
            }
        }
    };

#pragma unroll
    for (int a = 0; a < GP; ++a)
#pragma unroll
        for (int r = 0; r < R; ++r) total[a][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue_weights(0, 0);
    load_consts(0, cc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const int gs = gs_begin + step / NL, l = step % NL;
        const bool last_layer = (l == NL - 1);
        // results of the previous group set go out now: their write latency hides under this step's MFMAs instead
        // of stalling the vmcnt(0) that publishes the next weight slice
        if (po_gs >= 0) { store_pending(); po_gs = -1; }
        if (l == 0) load_operands(gs);
        if (step + 1 < nsteps) {
            load_consts(step + 1, cn);
            if (!(EXP & 4) || step == 0) issue_weights(step + 1, buf ^ 1);
        }
        const bool second = (gs * GP + 1) * 16 < p.count;      // the second group of the set exists (block-uniform)
        // ---- convT_l(G_l) for 16 GP channels: K = 3 map quads x 9 taps ----
        f32x4 acc[GP][R];
#pragma unroll
        for (int a = 0; a < GP; ++a)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[a][r] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* wb = s_w + buf * G::kWG;
        if constexpr (PIPE) {
            // fragments of map quad q+1 are requested before the 27 GP R MFMAs of quad q are issued, so the LDS
            // latency hides under this wave's own MFMAs (left to itself the compiler reads each B value right
            // before the 3 MFMAs that use it and stalls on lgkmcnt(0) every ~100 cycles)
            float av[2][3][R + 2], bw[2][9][GP];
            auto load_quad = [&](int quad, int set) {
                const float* a_base = s_g + (l * 12 + quad * 4 + lk) * G::kCS + wy * G::kCols + wx + li + G::kColOff;
                const float* b_base = wb + (quad * 4 + lk) * G::kNB + li;
#pragma unroll
                for (int r = 0; r < R + 2; ++r)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) av[set][dx][r] = a_base[r * G::kCols + dx];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int a = 0; a < GP; ++a) bw[set][tap][a] = b_base[tap * 12 * G::kNB + a * 16];
            };
            load_quad(0, 0);
#pragma unroll
            for (int quad = 0; quad < 3; ++quad) {
                const int set = quad & 1;
                if (quad + 1 < 3) load_quad(quad + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);        // keep the reads above, the MFMAs below
                if constexpr (BF != 0) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        bf16x4_bits ap[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) ap[r] = pack_bf16x4(av[set][dx][r], av[set][dx][r + 1], av[set][dx][r + 2], 0.f);
#pragma unroll
                        for (int a = 0; a < GP; ++a)
                            if (a == 0 || second) {
                                const bf16x4_bits bp = pack_bf16x4(bw[set][dx][a], bw[set][3 + dx][a], bw[set][6 + dx][a], 0.f);
#pragma unroll
                                for (int r = 0; r < R; ++r) acc[a][r] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ap[r], bp, acc[a][r], 0, 0, 0);
                            }
                    }
                } else
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int a = 0; a < GP; ++a)
                            if (a == 0 || second) {
#pragma unroll
                                for (int r = 0; r < R; ++r)
                                    acc[a][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[set][dx][r + dy], bw[set][dy * 3 + dx][a], acc[a][r], 0, 0, 0);
                            }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int quad = 0; quad < 3; ++quad) {
            const float* a_base = s_g + (l * 12 + quad * 4 + lk) * G::kCS + wy * G::kCols + wx + li + G::kColOff;
            const float* b_base = wb + (quad * 4 + lk) * G::kNB + li;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                float av[R + 2];
#pragma unroll
                for (int r = 0; r < R + 2; ++r) av[r] = a_base[r * G::kCols + dx];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
                    for (int a = 0; a < GP; ++a) {
                        if (a == 0 || second) {
                            const float b = b_base[(dy * 3 + dx) * 12 * G::kNB + a * 16];
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                acc[a][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r + dy], b, acc[a][r], 0, 0, 0);
                        }
                    }
                }
            }
        }
        }
        // ---- layer l's ReLU mask + BN backward, accumulated over the layers of the block ----
#pragma unroll
        for (int a = 0; a < GP; ++a) {
            const int co = (gs * GP + a) * 16 + li;
            const float scale = cc[a][0], beta = cc[a][1], mean = cc[a][2], rstd = cc[a][3];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = y0 + wy + r;
                if constexpr ((EXP & 32) != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) total[a][r][e] += acc[a][r][e];
                } else if (co < p.count && y < p.h && px + 3 < p.w) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xcen = xc[a][r][e] - mean;
                        const float z = fmaf(xcen, scale, beta);
                        const float dz = z > 0.f ? acc[a][r][e] : 0.f;
                        s1 += dz;
                        s2 += dz * (xcen * rstd);
                        total[a][r][e] += scale * dz;
                    }
                }
                if (last_layer) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) po[a][r][e] = dc[a][r][e] + total[a][r][e];
                    total[a][r] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (!(EXP & 8) && lk == 0) {
                float* red = s_red + buf * G::kRed + a * (4 * 16 * 2);
                red[(wave * 16 + li) * 2] = s1;
                red[(wave * 16 + li) * 2 + 1] = s2;
            }
        }
        if (last_layer) po_gs = gs;
#pragma unroll
        for (int a = 0; a < GP; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) cc[a][i] = cn[a][i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!(EXP & 8) && tid < 32 * GP) {
            const int a = tid >> 5, j = (tid >> 1) & 15, which = tid & 1;
            const int co = (gs * GP + a) * 16 + j;
            if (co < p.count) {
                const float* red = s_red + buf * G::kRed + a * (4 * 16 * 2);
                double t = 0.0;
                for (int wv = 0; wv < 4; ++wv) t += static_cast<double>(red[(wv * 16 + j) * 2 + which]);
                atomicAdd(p.scratch[l] + bn_slot_offset(p.slot_stride) + grp_off / 2 + 2 * co + which, t);
            }
        }
    }
    if (po_gs >= 0) store_pending();
}

template <int NL, int WX, int R, int GP = 1, int EXP = 0, int PIPE = 1, int VEC = 1, int BF = 0>
inline int launch_dgrad_block(DgradBlockParams p, hipStream_t stream) {
    using G = DgradBlockGeom<NL, WX, R, GP, VEC>;
    p.tiles_x = (p.w + G::kTileX - 1) / G::kTileX;
    const int tiles_y = (p.h + G::kTileY - 1) / G::kTileY;
    static bool configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& configured = configured_by_device[dev & 15];
    if (!configured && G::kBytes > 48 * 1024) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_block_kernel<NL, WX, R, GP, EXP, PIPE, VEC, BF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(G::kBytes)));
        configured = true;
    }
    // enough blocks for ~3 per CU: slice the channel group sets when the level has few tiles
    const int tiles = p.tiles_x * tiles_y * p.n;
    const int gsets = ((p.count + 15) / 16 + GP - 1) / GP;
    int ysplit = (768 + tiles - 1) / tiles;
    if (ysplit > gsets) ysplit = gsets;
    if (ysplit < 1) ysplit = 1;
    dgrad_block_kernel<NL, WX, R, GP, EXP, PIPE, VEC, BF><<<dim3(p.tiles_x * tiles_y, ysplit, p.n), kConvThreads, G::kBytes, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// 8-wave variant: the dY tile (52 KB for 4 layers) is what limits a CU to two blocks; here a block is 512 threads and
// its two halves (4 waves each, the same 32 x 6 pixel tile) work on DIFFERENT 16-channel groups of every step, each
// half with its own double-buffered weight slice.  Same LDS per CU, twice the waves per SIMD to hide each other's
// LDS waits, epilogues and barriers -- MFMA utilisation follows waves per SIMD on this part.
// Needs <= 128 registers per lane for the four waves per SIMD: no deferred stores, no register double-buffering.
// ---------------------------------------------------------------------------------------------
template <int NL>
struct DgradBlock8Geom {
    static constexpr int WX = 2, R = 3;
    static constexpr int kThreads = 512;
    static constexpr int kTileX = 32, kTileY = 6;
    static constexpr int kRows = kTileY + 2, kCols = kTileX + 2;
    static constexpr int kPlane = kRows * kCols;                       // 272
    static constexpr int kCS = ((kPlane - 16 + 31) / 32) * 32 + 16;    // 272
    static constexpr int kWG = 9 * 12 * 16;
    static constexpr int kWPre = (kWG + 255) / 256;                    // per thread of a half
    static constexpr int kRed = 4 * 16 * 2;                            // per (buffer, half): [4 waves][16][2]
    static constexpr size_t kBytes = sizeof(float) * (NL * 12 * kCS + 2 * 2 * kWG + 2 * 2 * kRed);
    static_assert(kBytes <= 80 * 1024, "two blocks per CU");
};

template <int NL, int BF = 0>
__global__ void __launch_bounds__(512, 4) dgrad_block8_kernel(const DgradBlockParams p0) {
    using G = DgradBlock8Geom<NL>;
    constexpr int R = G::R;
    const int grp = p0.group_n > 0 ? blockIdx.z / p0.group_n : 0;
    const int n = blockIdx.z - grp * p0.group_n;
    const DgradBlockParams& p = p0;
    const int64_t grp_off = grp * p0.gs;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_g = smem;                               // [NL*12][kCS]
    float* s_w = s_g + NL * 12 * G::kCS;             // [half][2][9][12][16]
    float* s_red = s_w + 4 * G::kWG;                 // [2][half][4 waves][16][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >> 2, w4 = wave & 3;
    const int th = tid & 255;                        // thread inside its half
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int tile = (gridDim.x & 7) == 0 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int x0 = (tile % p.tiles_x) * G::kTileX;
    const int y0 = (tile / p.tiles_x) * G::kTileY;
    const int wx = (w4 & 1) * 16;
    const int wy = (w4 >> 1) * R;
    const int px = x0 + wx + 4 * lk;
    // blockIdx.y: slice of the channel groups; inside the slice the two halves take alternate groups
    const int ngroups = (p.count + 15) / 16;
    const int g_per = (ngroups + gridDim.y - 1) / gridDim.y;
    const int g_begin = blockIdx.y * g_per;
    const int g_end = min(ngroups, g_begin + g_per);
    if (g_begin >= g_end) return;
    const int npairs = (g_end - g_begin + 1) / 2;
    const int nsteps = npairs * NL;

    // ---- G tiles: NL*12 maps with a 1-pixel halo, dword DMA by all 8 waves (once per block) ----
    {
        int goff = 0;
        bool ok = false;
        if (tid < G::kPlane) {
            const int ry = tid / G::kCols, rx = tid - ry * G::kCols;
            const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
            if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) { ok = true; goff = gy * p.g_w + gx; }
        }
        const float* g_n = p.g + grp_off + n * p.g_ns;
        const int e0 = wave * 64;
        if (e0 < G::kPlane) {
            for (int c = 0; c < NL * 12; ++c) {
                const float* src = ok ? g_n + static_cast<int64_t>(c) * p.g_cs + goff : g_pad_consts + 4;
                if (e0 + lane < G::kPlane) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_g + c * G::kCS + e0), 4, 0, 0);
            }
        }
    }

    // ---- this half's weight slice of a step: element (tap, c, j) <- W_l[c][w_ci_off + 16 group + j][8 - tap] ----
    int wc[G::kWPre], wrest[G::kWPre];
#pragma unroll
    for (int k = 0; k < G::kWPre; ++k) {
        const int e = th + k * 256;
        const int j = e % 16;
        const int rest = e / 16;
        const int c = rest % 12;
        const int tap = rest / 12;
        wc[k] = c * 9;
        wrest[k] = j * 9 + (8 - tap);
    }
    auto step_group = [&](int step) { return g_begin + 2 * (step / NL) + half; };
    auto issue_weights = [&](int step, int buf) {
        const int gq = step_group(step), l = step % NL;
        if (gq >= g_end) return;                                     // half-uniform
        const float* wl = p.wgt[l] + static_cast<int64_t>(p.w_ci_off + gq * 16) * 9;
        const int wcin = p.w_cin[l];
        float* dst = s_w + (half * 2 + buf) * G::kWG;
#pragma unroll
        for (int k = 0; k < G::kWPre; ++k) {
            const int e0 = k * 256 + w4 * 64;
            if (e0 < G::kWG) {
                const bool ok = gq * 16 + ((e0 + lane) & 15) < p.count;
                const float* src = ok ? wl + wc[k] * wcin + wrest[k] : g_pad_consts + 4;
                if (e0 + lane < G::kWG) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + e0), 4, 0, 0);
            }
        }
    };

    const float* x_n = p.x + grp_off + n * p.ns;
    float* out_n = p.out + grp_off + n * p.ns;
    f32x4 xc[R], dc[R], total[R];
#pragma unroll
    for (int r = 0; r < R; ++r) total[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue_weights(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const int gq = step_group(step), l = step % NL;
        const bool active = gq < g_end;                              // half-uniform
        const bool last_layer = (l == NL - 1);
        const int co = gq * 16 + li;
        float scale = 0.f, beta = 0.f, mean = 0.f, rstd = 0.f;
        if (active) {
            if (l == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int y = y0 + wy + r;
                    xc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                    dc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (co < p.count && y < p.h && px + 3 < p.w) {
                        xc[r] = *reinterpret_cast<const f32x4*>(x_n + static_cast<int64_t>(co) * p.cs + y * p.w + px);
                        if (co >= p.acc_from) dc[r] = *reinterpret_cast<const f32x4*>(out_n + static_cast<int64_t>(co) * p.cs + y * p.w + px);
                    }
                }
            }
            if (co < p.count) {
                mean = p.saved[l][grp_off + 2 * co]; rstd = p.saved[l][grp_off + 2 * co + 1];
                scale = p.gamma[l][co] * rstd;
                beta = p.beta[l][co];
            }
        }
        if (step + 1 < nsteps) issue_weights(step + 1, buf ^ 1);
        if (active) {
            // ---- convT_l(G_l) for this half's 16 channels: K = 3 map quads x 9 taps ----
            f32x4 acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* wb = s_w + (half * 2 + buf) * G::kWG;
#pragma unroll
            for (int quad = 0; quad < 3; ++quad) {
                const float* a_base = s_g + (l * 12 + quad * 4 + lk) * G::kCS + wy * G::kCols + wx + li;
                const float* b_base = wb + (quad * 4 + lk) * 16 + li;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float av[R + 2];
#pragma unroll
                    for (int r = 0; r < R + 2; ++r) av[r] = a_base[r * G::kCols + dx];
                    if constexpr (BF != 0) {
                        const bf16x4_bits bp = pack_bf16x4(b_base[dx * 12 * 16], b_base[(3 + dx) * 12 * 16], b_base[(6 + dx) * 12 * 16], 0.f);
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            acc[r] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pack_bf16x4(av[r], av[r + 1], av[r + 2], 0.f), bp, acc[r], 0, 0, 0);
                        continue;
                    }
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const float b = b_base[(dy * 3 + dx) * 12 * 16];
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r + dy], b, acc[r], 0, 0, 0);
                    }
                }
            }
            // ---- layer l's ReLU mask + BN backward, accumulated over the layers of the block ----
            // two pixels per instruction (v_pk_add / v_pk_fma): 9 VALU per pair; sum dz * xhat is kept as sum dz * (x - mean)
            // and scaled by rstd once
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 mean2 = {mean, mean}, scale2 = {scale, scale}, beta2 = {beta, beta};
            f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = y0 + wy + r;
                if (co < p.count && y < p.h && px + 3 < p.w) {
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) {
                        const f32x2 xv = {xc[r][2 * hp], xc[r][2 * hp + 1]};
                        const f32x2 xcen = xv - mean2;
                        const f32x2 z = __builtin_elementwise_fma(xcen, scale2, beta2);
                        f32x2 dz;
                        dz[0] = z[0] > 0.f ? acc[r][2 * hp] : 0.f;
                        dz[1] = z[1] > 0.f ? acc[r][2 * hp + 1] : 0.f;
                        s1v += dz;
                        s2v = __builtin_elementwise_fma(dz, xcen, s2v);
                        f32x2 tv = {total[r][2 * hp], total[r][2 * hp + 1]};
                        tv = __builtin_elementwise_fma(dz, scale2, tv);
                        total[r][2 * hp] = tv[0];
                        total[r][2 * hp + 1] = tv[1];
                    }
                    if (last_layer) {
                        f32x4 o = dc[r];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += total[r][e];
                        *reinterpret_cast<f32x4*>(out_n + static_cast<int64_t>(co) * p.cs + y * p.w + px) = o;
                    }
                }
                if (last_layer) total[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            float s1 = s1v[0] + s1v[1];
            float s2 = (s2v[0] + s2v[1]) * rstd;
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lk == 0) {
                float* red = s_red + (buf * 2 + half) * G::kRed;
                red[(w4 * 16 + li) * 2] = s1;
                red[(w4 * 16 + li) * 2 + 1] = s2;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (active && th < 32) {
            const int j = th >> 1, which = th & 1;
            const int cj = gq * 16 + j;
            if (cj < p.count) {
                const float* red = s_red + (buf * 2 + half) * G::kRed;
                double t = 0.0;
                for (int wv = 0; wv < 4; ++wv) t += static_cast<double>(red[(wv * 16 + j) * 2 + which]);
                atomicAdd(p.scratch[l] + bn_slot_offset(p.slot_stride) + grp_off / 2 + 2 * cj + which, t);
            }
        }
    }
}

template <int NL, int BF = 0>
inline int launch_dgrad_block8(DgradBlockParams p, hipStream_t stream) {
    using G = DgradBlock8Geom<NL>;
    p.tiles_x = (p.w + G::kTileX - 1) / G::kTileX;
    const int tiles_y = (p.h + G::kTileY - 1) / G::kTileY;
    static bool configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& configured = configured_by_device[dev & 15];
    if (!configured) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_block8_kernel<NL, BF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(G::kBytes)));
        configured = true;
    }
    const int tiles = p.tiles_x * tiles_y * p.n;
    const int pairs = ((p.count + 15) / 16 + 1) / 2;
    int ysplit = (1024 + tiles - 1) / tiles;          // 4 blocks per CU at the coarse levels (in-job A/B: -1.5 % on the family vs 512)
    if (ysplit > pairs) ysplit = pairs;
    if (ysplit < 1) ysplit = 1;
    dgrad_block8_kernel<NL, BF><<<dim3(p.tiles_x * tiles_y, ysplit, p.n), G::kThreads, G::kBytes, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// 16-byte DMA of the G tile: float4-aligned rows of the gradient maps
inline bool dgrad_block_vec_ok(const DgradBlockParams& p) {
    return (p.g_w % 4 == 0) && (p.g_cs % 4 == 0) && (p.g_ns % 4 == 0) && (p.gs % 4 == 0) && (reinterpret_cast<uintptr_t>(p.g) % 16 == 0);
}

// float4 epilogue: W % 4 == 0 and 16-byte aligned planes
inline bool dgrad_block_ok(const DgradBlockParams& p) {
    return (p.w % 4 == 0) && (p.cs % 4 == 0) && (p.ns % 4 == 0) && (reinterpret_cast<uintptr_t>(p.x) % 16 == 0) &&
           (reinterpret_cast<uintptr_t>(p.out) % 16 == 0);
}

}  // namespace endo
