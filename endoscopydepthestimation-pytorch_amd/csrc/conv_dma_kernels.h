// LDS-DMA variant of the implicit-GEMM convolution (same math, tiling, MFMA roles and epilogues as
// conv_kernels.h; see the header there).  What changes is how a K-chunk reaches LDS:
//
//   * `global_load_lds_dword`: every lane hands the DMA engine its own global address and the wave
//     writes 64 consecutive LDS dwords -- no staging VGPRs, no ds_write pass, no address VALU in the
//     loop.  Out-of-image / out-of-range lanes point at a 1-float pad constant instead of branching:
//     0.0 for raw inputs and weights, NaN for BN+ReLU inputs.
//   * BN+ReLU moves from the store path to the fragment read: a = max((raw - mean) * scale + beta, 0)
//     right after the ds_read.  max(NaN, 0) = 0, so the NaN pad is exactly the zero padding of the
//     post-activation tensor (reference models.py:22-25: the conv pads relu(bn(x)), not x).
//   * two LDS buffers, ONE barrier per chunk: wait own DMA(c) -> barrier -> issue DMA(c+1) into the
//     other buffer -> MFMAs on chunk c.  HBM latency hides under the MFMAs of the same block instead
//     of relying on a co-resident block being in a different phase.
//   * the ~30-60 staging registers are gone, so the growth-12 forward kernel fits 5 waves/SIMD and a
//     256x320x8 level-0 launch (1280 tiles) is exactly one round of the chip.
#pragma once

#include "conv_kernels.h"

namespace endo {

static __device__ __attribute__((aligned(16))) float g_pad_consts[8] = {__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""),
                                                                  0.0f, 0.0f, 0.0f, 0.0f};   // [0..3] NaN pad, [4..7] zero pad

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int KS, int KC, int Q, int WX, int R, int NBUF, int VEC, int IN = IN_PLAIN>
struct ConvDmaSmem {
    using G = ConvGeom<KS, KC, WX, R, VEC>;
    static constexpr int kW = KS * KS * KC * 16 * Q;
    // IN_UNPOOL stages the POOLED gradient tile plus its argmax codes (4 one-byte codes per dword) and expands
    // them when the A fragment is read; channel stride == 8 (mod 32): the 4 k-groups x 8 pooled columns of a
    // fragment read hit 32 different banks
    static constexpr int kUG = (G::kTileX / 2) * (G::kTileY / 2);
    static constexpr int kUI = kUG / 4;
    static constexpr int kUStride = ((kUG + kUI - 8 + 31) / 32) * 32 + 8;
    static constexpr int kChan = (IN == IN_UNPOOL) ? kUStride : G::kCS;
    static constexpr int kBuf = KC * kChan + kW;                  // floats per buffer
    static constexpr int kTail = 4 * 16 * Q + 8 * 16 * Q;       // dgrad constants + reduction scratch
    static size_t bytes(int bn_cap) { return sizeof(float) * (NBUF * kBuf + 3 * bn_cap + kTail); }
};

// MINW: minimum waves per SIMD the register allocator must leave room for (__launch_bounds__'s
// second argument on AMD); 5 for the growth-12 forward kernel so that 5 blocks share a CU.
// VEC: 1 = dword DMA (any shape), 4 = 16-byte DMA (W % 4 == 0, PLAIN / BNRELU inputs): 4x fewer DMA
// instructions and address computations per chunk.
// XF: where BN+ReLU is applied.  0 = on every A-fragment read (3 VALU per read, each LDS value is
// read by 3 taps); 1 = once, in place in LDS, by the thread whose DMA wrote the value, right after
// its own vmcnt(0) and before the barrier that publishes the chunk (no extra synchronisation).
// EXP: diagnostic bit mask for tools/conv_bench (0 in the library): 1 = only the first chunk is DMA'd, 2 = no BN+ReLU on the fragment read, 4 = DMA never waited for (racy: timing only)
// PH: sub-pixel row phase of the transition-up forward (-1 = ordinary convolution).  nearest x2 followed by a 3x3
// convolution (reference models.py:70-80) is, for the output pixels (2y+a, 2x+b), a 2x2 convolution of the LOW-resolution
// input with tap-summed weights: the 3 upsampled rows a 3x3 window covers are only 2 distinct input rows.  With PH = a the
// kernel runs on the low-resolution grid, its 16*Q "output channels" are [b = 0 | b = 1] x Q/2 tiles of real channels
// (weights pre-summed by tu_phase_weights_kernel), the taps a phase does not use are skipped at compile time (16 of 36
// tap-phase pairs remain: 4/9 of the MACs) and the epilogue interleaves the two column phases into full-resolution rows.
// BF: 1 = bf16 MFMA operands (ENDO_OPT_MFMA_BF16; 3x3, ordinary inputs only): the three row taps (dy = 0..2) of one input channel
// and column tap are the k = 4 lk .. 4 lk + 2 of ONE v_mfma_f32_16x16x16_bf16 (the fourth k is a zero), so a channel quad costs
// 3 of those instead of 9 v_mfma_f32_16x16x4_f32; operands are rounded to bf16 after BN + ReLU, accumulation stays fp32.
template <int KS, int KC, int Q, int IN, int EPI, int WX, int R, int NBUF, int MINW, int VEC, int XF, int EXP = 0, int PH = -1, int BF = 0>
__global__ void __launch_bounds__(kConvThreads, MINW) conv_dma_kernel(const ConvParams p0) {
    static_assert(BF == 0 || (PH < 0 && (KS == 3 || KS == 1) && (IN == IN_BNRELU || IN == IN_PLAIN || (IN == IN_UNPOOL && KS == 1))),
                  "bf16 operands: the ordinary 3x3 convolution and the 1x1 convolutions of the transition-down layers");
    static_assert(PH < 0 || (KS == 3 && (Q % 2) == 0 && IN == IN_PLAIN && EPI == EPI_FWD), "phase mode is the transition-up forward");
    static_assert(IN != IN_UNPOOL || (KS == 1 && R % 2 == 0), "UNPOOL is the transition-down data gradient (1x1)");
    static_assert(NBUF == 1 || NBUF == 2, "one or two LDS buffers");      // (three buffers, DMA two chunks ahead: 7 % slower in the in-job A/B)
    static_assert(VEC == 1 || (VEC == 4 && IN != IN_UPSAMPLE && IN != IN_UNPOOL && IN != IN_SUBPIX), "16-byte DMA needs contiguous sources");
    // IN_SUBPIX: the transition-up data gradient in sub-pixel form.  d(low-res input) = sum over the four sub-pixel phases
    // (alpha, beta) of the full-resolution dY, each a stride-2 sub-sampled plane ("pseudo input channel" (alpha, beta, co))
    // seen through 2 x 2 of the 3 x 3 low-resolution taps with tap-summed weights: 16 instead of 36 tap-phase pairs.
    static_assert(IN != IN_SUBPIX || (KS == 3 && EPI == EPI_FWD && PH < 0), "sub-pixel input mode is the transition-up data gradient");
    using G = ConvGeom<KS, KC, WX, R, VEC>;
    using S = ConvDmaSmem<KS, KC, Q, WX, R, NBUF, VEC, IN>;
    constexpr int KK = KS * KS;
    constexpr int NB = 16 * Q;
    constexpr int kWElems = S::kW;
    constexpr int kWPre = (kWElems + kConvThreads - 1) / kConvThreads;
    constexpr bool kDgrad = (EPI == EPI_DGRAD_BN || EPI == EPI_DGRAD_SUMPOOL);
    constexpr int kPhW = 2 * 12 * 16 + 16;          // phase mode: floats per input channel (== 16 mod 32: conflict-free B reads)
    static_assert(PH < 0 || KC * kPhW <= kWElems, "the phase weights fit the ordinary weight buffer");
    constexpr int kSubW = 4 * 16 * Q + 16;          // IN_SUBPIX: floats per pseudo input channel (== 16 mod 32 for Q = 3)

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_aux = smem + NBUF * S::kBuf;
    int grp, n;
    group_of(p0, blockIdx.z, grp, n);
    const ConvParams p = group_view(p0, grp);
    const int groups = p0.group_n > 0 ? gridDim.z / p0.group_n : 1;
    const bool first_of_group = blockIdx.x == 0 && blockIdx.y == 0 && n == 0;
    (void)groups; (void)first_of_group;
    const int cap = p.bn_cap;      // BN tables: scale [0,cap) mean [cap,2cap) beta [2cap,3cap); then dgrad constants, reductions

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blockIdx.x is the fastest-varying dispatch index; when the tile count is a multiple of 8 the XCD of a
    // block is blockIdx.x % 8 for every (y, z) slice, and the remap keeps neighbouring tiles on one XCD
    const int tile = (gridDim.x & 7) == 0 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int x0 = (tile % p.tiles_x) * G::kTileX;
    const int y0 = (tile / p.tiles_x) * G::kTileY;
    // PH == 2: both row phases of the transition-up forward in ONE launch, the block's phase = blockIdx.y (round 6: at the coarse levels a phase
    // is 32 - 384 blocks, and two under-filled launches of 27 us each cost what one does)
    const int ph = PH == 2 ? static_cast<int>(blockIdx.y) : PH;
    const int co_base = (p.ksplit > 0 || PH == 2) ? 0 : blockIdx.y * NB;

    // ---------------- prologue: per-channel constants (identical to conv_mfma_kernel) ----------------
    if constexpr (IN == IN_BNRELU) {
        for (int c = tid; c < p.cin; c += kConvThreads) {
            float scale, mean, beta;
            bn_input_constants(p, p0, grp, groups, first_of_group, c, scale, mean, beta);
            s_aux[c] = scale;
            s_aux[cap + c] = mean;
            s_aux[2 * cap + c] = beta;
        }
        // channels past cin (last chunk): any finite constants; their raw values are the NaN pad
        for (int c = p.cin + tid; c < ((p.cin + KC - 1) / KC) * KC; c += kConvThreads) {
            s_aux[c] = 0.f; s_aux[cap + c] = 0.f; s_aux[2 * cap + c] = 0.f;
        }
    }
    if constexpr (EPI == EPI_DGRAD_BN) {
        float* cst = s_aux + 3 * cap;
        if (tid < NB) {
            const int c = co_base + tid;
            float mean = 0.f, rstd = 0.f, scale = 0.f, beta = 0.f;
            if (c < p.cout) {
                mean = p.bn_saved[2 * c];
                rstd = p.bn_saved[2 * c + 1];
                scale = p.bn_gamma[c] * rstd;
                beta = p.bn_beta[c];
            }
            cst[4 * tid] = scale; cst[4 * tid + 1] = beta; cst[4 * tid + 2] = mean; cst[4 * tid + 3] = rstd;
        }
    }

    f32x4 acc[R][Q];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wx = (wave % WX) * 16;
    const int wy = (wave / WX) * R;
    const int li = lane & 15;
    const int lk = lane >> 4;

    // this thread's fixed tile positions (units of VEC floats): global offset inside a channel plane, or "pad"
    int goff[G::kPos];
    unsigned pos_ok = 0;
#pragma unroll
    for (int k = 0; k < G::kPos; ++k) {
        const int e = tid + k * kConvThreads;
        goff[k] = 0;
        if (e < G::kUnits) {
            const int ry = e / (G::kCols / VEC);
            const int rx = (e - ry * (G::kCols / VEC)) * VEC;
            const int gy = y0 - G::kHalo + ry;
            const int gx = x0 - G::kLeft + rx;
            if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) {      // W % VEC == 0: a unit is entirely in or out
                pos_ok |= (1u << k);
                if constexpr (IN == IN_UPSAMPLE) goff[k] = (gy >> 1) * p.in_w + (gx >> 1);
                else if constexpr (IN == IN_SUBPIX) goff[k] = 2 * gy * p.in_w + 2 * gx;
                else goff[k] = gy * p.in_w + gx;
            }
        }
    }
    // IN_UNPOOL: this thread's slots of the [KC][kUStride] staging layout: pooled-plane offset (floats for the
    // gradient, bytes for the codes; -1 = padding / out of image) and which array it comes from
    constexpr int kUSlots = (IN == IN_UNPOOL) ? (KC * S::kUStride + kConvThreads - 1) / kConvThreads : 1;
    int uoff[kUSlots];
    if constexpr (IN == IN_UNPOOL) {
        constexpr int PC = G::kTileX / 2;
        const int ph = p.h >> 1, pw = p.w >> 1;
#pragma unroll
        for (int k = 0; k < kUSlots; ++k) {
            const int e = tid + k * kConvThreads;
            const int u = e % S::kUStride;
            int off = -1;
            if (e < KC * S::kUStride) {
                if (u < S::kUG) {
                    const int gy = (y0 >> 1) + u / PC, gx = (x0 >> 1) + u % PC;
                    if (gy < ph && gx < pw) off = gy * p.in_w + gx;
                } else if (u < S::kUG + S::kUI) {
                    const int q = u - S::kUG;
                    const int gy = (y0 >> 1) + q / (PC / 4), gx = (x0 >> 1) + 4 * (q % (PC / 4));
                    if (gy < ph && gx < pw) off = gy * p.in_w + gx;       // pooled width % 4 == 0: a code dword is all in or all out
                }
            }
            uoff[k] = off;
        }
    }
    const float* in_n = p.in + n * p.in_ns;
    const float* pad_in = g_pad_consts + (IN == IN_BNRELU ? 0 : 4);
    const float* pad_zero = g_pad_consts + 4;
    const int nchunks_all = (p.cin + KC - 1) / KC;
    int chunk_begin = 0, nchunks = nchunks_all;
    ConvParams po = p;
    if (p.ksplit > 0) {          // this block's slice of the K loop
        const int per = (nchunks_all + p.ksplit - 1) / p.ksplit;
        chunk_begin = blockIdx.y * per;
        nchunks = min(nchunks_all, chunk_begin + per);
        po.out = p.out + blockIdx.y * p.split_stride;
    }

    auto issue_dma = [&](int chunk, int buf) {
        const int c_base = chunk * KC;
        float* s_in = smem + buf * S::kBuf;
        float* s_w = s_in + KC * S::kChan;
        if constexpr (IN == IN_UNPOOL) {
            const uint8_t* idx_n = p.in_idx + n * p.idx_ns;
#pragma unroll
            for (int k = 0; k < kUSlots; ++k) {
                const int e0 = k * kConvThreads + wave * 64;
                if (e0 < KC * S::kUStride) {
                    const int e = e0 + lane;
                    const int ch = c_base + e / S::kUStride;
                    const bool is_code = (e % S::kUStride) >= S::kUG;
                    const void* src = pad_zero;
                    if (uoff[k] >= 0 && ch < p.cin) {
                        const int64_t o = static_cast<int64_t>(ch) * p.in_cs + uoff[k];
                        src = is_code ? static_cast<const void*>(idx_n + o) : static_cast<const void*>(in_n + o);
                    }
                    if (e < KC * S::kUStride) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_in + e0), 4, 0, 0);
                }
            }
        }
        int sub_ph = 0;          // IN_SUBPIX: the chunk's sub-pixel phase (a chunk never straddles phases: KC divides sub_c)
        if constexpr (IN == IN_SUBPIX) sub_ph = c_base / p.sub_c;
#pragma unroll
        for (int c = 0; c < (IN == IN_UNPOOL ? 0 : KC); ++c) {
            const int ch = c_base + c;
            const float* plane = in_n + static_cast<int64_t>(ch) * p.in_cs;
            if constexpr (IN == IN_SUBPIX)
                plane = in_n + static_cast<int64_t>(ch - sub_ph * p.sub_c) * p.in_cs + (sub_ph >> 1) * p.in_w + (sub_ph & 1);
#pragma unroll
            for (int k = 0; k < G::kPos; ++k) {
                const int e0 = k * kConvThreads + wave * 64;          // wave-uniform first unit
                if (e0 < G::kUnits) {
                    const bool ok = (ch < p.cin) && (pos_ok & (1u << k));
                    const float* src = ok ? plane + goff[k] : pad_in;
                    if (e0 + lane < G::kUnits) {
                        if constexpr (VEC == 4)
                            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_in + c * G::kCS + 4 * e0), 16, 0, 0);
                        else
                            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_in + c * G::kCS + e0), 4, 0, 0);
                    }
                }
            }
        }
        if constexpr (IN == IN_SUBPIX) {
            // compact weights: [pseudo channel][kSubW] floats = 2 x 2 taps x NB output channels (+ pad)
            constexpr int kUnitsW = KC * kSubW / 4;
            const float* wsrc = p.wgt + static_cast<int64_t>(c_base) * kSubW;
#pragma unroll
            for (int k = 0; k < (kUnitsW + kConvThreads - 1) / kConvThreads; ++k) {
                const int u0 = k * kConvThreads + wave * 64;
                if (u0 < kUnitsW && u0 + lane < kUnitsW)
                    __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + 4 * (u0 + lane)), (lptr_t)(s_w + 4 * u0), 16, 0, 0);
            }
        }
        if constexpr (PH >= 0) {
            // compact phase weights: [ci][kPhW] floats, the 2 x 12 (row tap, column-tap x tile) slices a row phase uses
            constexpr int kUnitsW = KC * kPhW / 4;
            const float* wsrc = p.wgt + static_cast<int64_t>(c_base) * kPhW + (PH == 2 ? static_cast<int64_t>(ph) * p.cin * 400 : 0);          // (tu_phase_weights_kernel: phase 1's slices 400 floats per input channel further on)
#pragma unroll
            for (int k = 0; k < (kUnitsW + kConvThreads - 1) / kConvThreads; ++k) {
                const int u0 = k * kConvThreads + wave * 64;
                if (u0 < kUnitsW && u0 + lane < kUnitsW)
                    __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + 4 * (u0 + lane)), (lptr_t)(s_w + 4 * u0), 16, 0, 0);
            }
        }
        constexpr bool kChunked = KS == 3 && KC == 16 && Q == 1 && EPI == EPI_FWD && PH < 0 && IN != IN_SUBPIX && BF == 0;
        bool gather = !(PH >= 0 || IN == IN_SUBPIX);
        if constexpr (kChunked) {
            if (p.wgt_chunks) {          // (block-uniform) the chunk's slice is one contiguous run in LDS order: 16-byte units, no index arithmetic
                gather = false;
                constexpr int kUnitsW = kWElems / 4;
                const float* wsrc = p.wgt_chunks + static_cast<int64_t>(chunk) * kWElems;
#pragma unroll
                for (int k = 0; k < (kUnitsW + kConvThreads - 1) / kConvThreads; ++k) {
                    const int u0 = k * kConvThreads + wave * 64;
                    if (u0 < kUnitsW && u0 + lane < kUnitsW)
                        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + 4 * (u0 + lane)), (lptr_t)(s_w + 4 * u0), 16, 0, 0);
                }
            }
        }
        if (gather)
#pragma unroll
        for (int k = 0; k < ((PH >= 0 || IN == IN_SUBPIX) ? 0 : kWPre); ++k) {
            const int e0 = k * kConvThreads + wave * 64;
            if (e0 < kWElems) {
                const int e = e0 + lane;
                const int j = e % NB;
                const int rest = e / NB;
                const int c = rest % KC;
                const int tap = rest / KC;
                const int ci = c_base + c;
                const int co = co_base + j;
                const float* src = pad_zero;
                if (e < kWElems && ci < p.cin && co < (PH >= 0 ? p.w_cout : p.cout)) {
                    if constexpr (kDgrad) src = p.wgt + (static_cast<int64_t>(ci) * p.w_cin + co) * KK + (KK - 1 - tap);
                    else src = p.wgt + (static_cast<int64_t>(co) * p.w_cin + ci) * KK + tap;
                }
                if (e < kWElems) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_w + e0), 4, 0, 0);
            }
        }
    };

    // DGRAD_BN: the epilogue's operands (activations x, and the gradient buffer it accumulates into)
    // do not depend on the MFMAs, so their loads are issued now and fly under the whole K loop.
    constexpr int PR = (EPI == EPI_DGRAD_BN) ? R : 1, PQ = (EPI == EPI_DGRAD_BN) ? Q : 1;
    f32x4 xpre[PR][PQ], dpre[PR][PQ];
    bool use_pre = false;
    if constexpr (EPI == EPI_DGRAD_BN) {
        use_pre = ((p.out_w & 3) == 0) && ((p.w & 3) == 0);
        if (use_pre) {
            const int px = x0 + wx + 4 * lk;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int co = co_base + q * 16 + li;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int y = y0 + wy + r;
                    xpre[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    dpre[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (co < p.cout && y < p.h && px + 3 < p.w) {
                        xpre[r][q] = *reinterpret_cast<const f32x4*>(p.x + n * p.x_ns + static_cast<int64_t>(co) * p.x_cs + y * p.out_w + px);
                        if (co >= p.acc_from)
                            dpre[r][q] = *reinterpret_cast<const f32x4*>(p.out + n * p.out_ns + static_cast<int64_t>(co) * p.out_cs + y * p.out_w + px);
                    }
                }
            }
        }
    }

    constexpr bool kInPlace = (IN == IN_BNRELU) && (XF == 1);
    auto transform_own = [&](int chunk, int buf) {
        if constexpr (kInPlace) {
            float* s_in = smem + buf * S::kBuf;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int ch = chunk * KC + c;
                const float sc = s_aux[ch], mn = s_aux[cap + ch], bt = s_aux[2 * cap + ch];
#pragma unroll
                for (int k = 0; k < G::kPos; ++k) {
                    const int e = tid + k * kConvThreads;
                    if (e < G::kUnits) {
                        if constexpr (VEC == 4) {
                            f32x4* q = reinterpret_cast<f32x4*>(s_in + c * G::kCS + 4 * e);
                            f32x4 v = *q;
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaxf(fmaf(v[i] - mn, sc, bt), 0.f);
                            *q = v;
                        } else {
                            float* q = s_in + c * G::kCS + e;
                            *q = __builtin_fmaxf(fmaf(*q - mn, sc, bt), 0.f);
                        }
                    }
                }
            }
        }
    };

    auto compute = [&](int chunk, int buf) {
        const float* s_in = smem + buf * S::kBuf;
        const float* s_w = s_in + KC * S::kChan;
        int sub_alpha = 0, sub_beta = 0;          // IN_SUBPIX: sub-pixel phase of this chunk's pseudo-channels (block-uniform)
        if constexpr (IN == IN_SUBPIX) {
            const int sub_ph = (chunk * KC) / p.sub_c;
            sub_alpha = sub_ph >> 1; sub_beta = sub_ph & 1;
        }
        (void)sub_alpha; (void)sub_beta;
        if constexpr (IN == IN_UNPOOL) {
            // a[row][x] = g[row/2][x/2] if code[row/2][x/2] == 2*(row&1) + (x&1) else 0 (y0, wy and wx are even)
            constexpr int PC = G::kTileX / 2;
            const int pcol = (wx + li) >> 1;
            const unsigned sh = 8u * (pcol & 3);
            const unsigned want = li & 1;
            if constexpr (BF != 0) {
                // 1x1: the channel quads of a chunk are the k = 4 lk + i of one bf16 instruction (missing quads are zeros)
                constexpr int NQ = KC / 4;
#pragma unroll
                for (int qg = 0; qg < (NQ + 3) / 4; ++qg) {
                    float a[4][R], b[4][Q];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int quad = qg * 4 + i;
                        if (quad < NQ) {
                            const float* g_base = s_in + (quad * 4 + lk) * S::kUStride + (wy >> 1) * PC + pcol;
                            const unsigned* i_base = reinterpret_cast<const unsigned*>(s_in + (quad * 4 + lk) * S::kUStride + S::kUG);
                            const float* b_base = s_w + (quad * 4 + lk) * NB + li;
#pragma unroll
                            for (int rr = 0; rr < R / 2; ++rr) {
                                const float g = g_base[rr * PC];
                                const unsigned code = (i_base[(((wy >> 1) + rr) * PC + pcol) >> 2] >> sh) & 0xffu;
                                a[i][2 * rr] = code == want ? g : 0.f;
                                a[i][2 * rr + 1] = code == want + 2u ? g : 0.f;
                            }
#pragma unroll
                            for (int q = 0; q < Q; ++q) b[i][q] = b_base[q * 16];
                        } else {
#pragma unroll
                            for (int r = 0; r < R; ++r) a[i][r] = 0.f;
#pragma unroll
                            for (int q = 0; q < Q; ++q) b[i][q] = 0.f;
                        }
                    }
                    bf16x4_bits ap[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) ap[r] = pack_bf16x4(a[0][r], a[1][r], a[2][r], a[3][r]);
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const bf16x4_bits bp = pack_bf16x4(b[0][q], b[1][q], b[2][q], b[3][q]);
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ap[r], bp, acc[r][q], 0, 0, 0);
                    }
                }
                return;
            }
#pragma unroll
            for (int quad = 0; quad < KC / 4; ++quad) {
                const float* g_base = s_in + (quad * 4 + lk) * S::kUStride + (wy >> 1) * PC + pcol;
                const unsigned* i_base = reinterpret_cast<const unsigned*>(s_in + (quad * 4 + lk) * S::kUStride + S::kUG);
                const float* b_base = s_w + (quad * 4 + lk) * NB + li;
                float a[R];
#pragma unroll
                for (int rr = 0; rr < R / 2; ++rr) {
                    const float g = g_base[rr * PC];
                    const unsigned code = (i_base[(((wy >> 1) + rr) * PC + pcol) >> 2] >> sh) & 0xffu;
                    a[2 * rr] = code == want ? g : 0.f;
                    a[2 * rr + 1] = code == want + 2u ? g : 0.f;
                }
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const float b = b_base[q * 16];
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b, acc[r][q], 0, 0, 0);
                }
            }
            return;
        }
        if constexpr (BF != 0 && KS == 1) {
            constexpr int NQ = KC / 4;
#pragma unroll
            for (int qg = 0; qg < (NQ + 3) / 4; ++qg) {
                float a[4][R], b[4][Q];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int quad = qg * 4 + i;
                    if (quad < NQ) {
                        const float* a_base = s_in + (quad * 4 + lk) * G::kCS + wy * G::kCols + wx + li + G::kColOff;
                        const float* b_base = s_w + (quad * 4 + lk) * NB + li;
                        float sc = 1.f, mn = 0.f, bt = 0.f;
                        if constexpr (IN == IN_BNRELU && !kInPlace) {
                            const int ch = chunk * KC + quad * 4 + lk;
                            sc = s_aux[ch]; mn = s_aux[cap + ch]; bt = s_aux[2 * cap + ch];
                        }
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const float v = a_base[r * G::kCols];
                            a[i][r] = (IN == IN_BNRELU && !kInPlace) ? __builtin_fmaxf(fmaf(v - mn, sc, bt), 0.f) : v;
                        }
#pragma unroll
                        for (int q = 0; q < Q; ++q) b[i][q] = b_base[q * 16];
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) a[i][r] = 0.f;
#pragma unroll
                        for (int q = 0; q < Q; ++q) b[i][q] = 0.f;
                    }
                }
                bf16x4_bits ap[R];
#pragma unroll
                for (int r = 0; r < R; ++r) ap[r] = pack_bf16x4(a[0][r], a[1][r], a[2][r], a[3][r]);
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const bf16x4_bits bp = pack_bf16x4(b[0][q], b[1][q], b[2][q], b[3][q]);
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ap[r], bp, acc[r][q], 0, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int quad = 0; quad < KC / 4; ++quad) {
            const float* a_base = s_in + (quad * 4 + lk) * G::kCS + wy * G::kCols + wx + li + G::kColOff;
            const float* b_base = s_w + (quad * 4 + lk) * NB + li;
            float sc = 1.f, mn = 0.f, bt = 0.f;
            if constexpr (IN == IN_BNRELU && !kInPlace) {
                const int ch = chunk * KC + quad * 4 + lk;
                sc = s_aux[ch]; mn = s_aux[cap + ch]; bt = s_aux[2 * cap + ch];
            }
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
                float a[R + KS - 1];
                if constexpr (BF != 0) {
#pragma unroll
                    for (int r = 0; r < R + 2; ++r) {
                        const float v = a_base[r * G::kCols + dx];
                        a[r] = (IN == IN_BNRELU && !kInPlace) ? __builtin_fmaxf(fmaf(v - mn, sc, bt), 0.f) : v;
                    }
                    bf16x4_bits ap[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) ap[r] = pack_bf16x4(a[r], a[r + 1], a[r + 2], 0.f);
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const bf16x4_bits bp = pack_bf16x4(b_base[(0 * KS + dx) * KC * NB + q * 16], b_base[(1 * KS + dx) * KC * NB + q * 16],
                                                           b_base[(2 * KS + dx) * KC * NB + q * 16], 0.f);
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ap[r], bp, acc[r][q], 0, 0, 0);
                    }
                    continue;
                }
                if constexpr (IN == IN_BNRELU && !kInPlace && !(EXP & 2)) {
                    // BN + ReLU two rows at a time: v_pk_add_f32 / v_pk_fma_f32 do the subtract and the fma of both values
                    // in one instruction each (same roundings as the scalar fmaf), so 4 VALU per pair instead of 6
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    constexpr int NR = R + KS - 1;
                    const f32x2 mn2 = {mn, mn}, sc2 = {sc, sc}, bt2 = {bt, bt};
#pragma unroll
                    for (int r = 0; r + 1 < NR; r += 2) {
                        f32x2 v = {a_base[r * G::kCols + dx], a_base[(r + 1) * G::kCols + dx]};
                        v = __builtin_elementwise_fma(v - mn2, sc2, bt2);
                        a[r] = __builtin_fmaxf(v[0], 0.f);
                        a[r + 1] = __builtin_fmaxf(v[1], 0.f);
                    }
                    if constexpr (NR & 1) a[NR - 1] = __builtin_fmaxf(fmaf(a_base[(NR - 1) * G::kCols + dx] - mn, sc, bt), 0.f);
                } else {
#pragma unroll
                    for (int r = 0; r < R + KS - 1; ++r) a[r] = a_base[r * G::kCols + dx];
                }
                if constexpr (IN == IN_SUBPIX) {
                    if (dx == (sub_beta == 0 ? 0 : 2)) continue;          // beta = 0 sees columns x, x+1; beta = 1 columns x-1, x
                }
#pragma unroll
                for (int dy = 0; dy < KS; ++dy) {
                    if (PH >= 0 && dy == (ph == 0 ? 2 : 0)) continue;          // row phase 0 sees input rows y-1, y; phase 1 rows y, y+1 (PH == 2: block-uniform at run time)
                    if constexpr (IN == IN_SUBPIX) {
                        if (dy == (sub_alpha == 0 ? 0 : 2)) continue;     // alpha = 0 sees rows y, y+1; alpha = 1 rows y-1, y
                    }
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        if (PH >= 0 && ((q < Q / 2 && dx == 2) || (q >= Q / 2 && dx == 0))) continue;      // column phase of this tile
                        float b;
                        if constexpr (IN == IN_SUBPIX) {
                            const int tyi = dy - (sub_alpha == 0 ? 1 : 0), txi = dx - (sub_beta == 0 ? 1 : 0);
                            b = s_w[(quad * 4 + lk) * kSubW + (tyi * 2 + txi) * NB + q * 16 + li];
                        } else if constexpr (PH >= 0) {
                            const int slot = dx == 0 ? q : (dx == 1 ? Q / 2 + q : Q / 2 + Q + (q - Q / 2));
                            b = s_w[(quad * 4 + lk) * kPhW + ((dy - ph) * 2 * Q + slot) * 16 + li];
                        } else {
                            b = b_base[(dy * KS + dx) * KC * NB + q * 16];
                        }
#pragma unroll
                        for (int r = 0; r < R; ++r)
                            acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r + dy], b, acc[r][q], 0, 0, 0);
                    }
                }
            }
        }
    };

    issue_dma(chunk_begin, 0);
    if constexpr (kInPlace) {
        // pipeline: [DMA(c+1) -> other buffer] [MFMAs(c)] [own DMA(c+1) landed -> transform own values] [barrier]
        __syncthreads();                              // s_aux (BN constants) visible
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        transform_own(chunk_begin, 0);
        __syncthreads();
        for (int chunk = chunk_begin; chunk < nchunks; ++chunk) {
            const int b = (chunk - chunk_begin) & 1;
            if (chunk + 1 < nchunks) issue_dma(chunk + 1, b ^ 1);
            compute(chunk, b);
            if (chunk + 1 < nchunks) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                transform_own(chunk + 1, b ^ 1);
            }
            __syncthreads();
        }
    } else {
        for (int chunk = chunk_begin; chunk < nchunks; ++chunk) {
            const int b = (chunk - chunk_begin) % NBUF;
            // own DMA of this chunk has landed; after the barrier everybody's has, and everybody has
            // finished reading the other buffer (chunk - 1), so it can be refilled
            if (!(EXP & 4) || chunk == chunk_begin) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (NBUF == 2 && chunk + 1 < nchunks && !(EXP & 1)) issue_dma(chunk + 1, b ^ 1);
            compute(chunk, b);
            if (NBUF == 1 && chunk + 1 < nchunks) {      // single buffer: refill only once everybody is done reading
                __syncthreads();
                issue_dma(chunk + 1, 0);
            }
        }
    }

    if constexpr (EPI == EPI_DGRAD_BN) {
        if (use_pre) {
            const float* cst = s_aux + 3 * cap;
            float* s_red = s_aux + 3 * cap + 4 * NB;
            const int px = x0 + wx + 4 * lk;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int jloc = q * 16 + li;
                const int co = co_base + jloc;
                const float scale = cst[4 * jloc], beta = cst[4 * jloc + 1], mean = cst[4 * jloc + 2], rstd = cst[4 * jloc + 3];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int y = y0 + wy + r;
                    if (co < p.cout && y < p.h && px + 3 < p.w) {
                        f32x4 o = dpre[r][q];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float xc = xpre[r][q][e] - mean;
                            const float z = fmaf(xc, scale, beta);
                            const float dz = z > 0.f ? acc[r][q][e] : 0.f;
                            s1 += dz;
                            s2 += dz * (xc * rstd);
                            o[e] += scale * dz;
                        }
                        *reinterpret_cast<f32x4*>(p.out + n * p.out_ns + static_cast<int64_t>(co) * p.out_cs + y * p.out_w + px) = o;
                    }
                }
                s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
                if (lk == 0) {
                    s_red[((tid >> 6) * NB + jloc) * 2] = s1;
                    s_red[((tid >> 6) * NB + jloc) * 2 + 1] = s2;
                }
            }
            __syncthreads();
            if (tid < 2 * NB) {
                const int j = tid >> 1, which = tid & 1;
                if (co_base + j < p.cout) {
                    double t = 0.0;
                    for (int wv = 0; wv < 4; ++wv) t += static_cast<double>(s_red[(wv * NB + j) * 2 + which]);
                    atomicAdd(p.bn_scratch + bn_slot_offset(p.bn_slot_stride) + 2 * (co_base + j) + which, t);
                }
            }
            return;
        }
    }
    if constexpr (PH >= 0) {
        // lane holds, for real channel co = q*16 + li (q < Q/2), low-resolution pixels x = px..px+3 of rows y0+wy+r in both
        // column phases (tiles q and q + Q/2): full-resolution row 2y + PH, columns 2px .. 2px+7 -- two float4 stores
        constexpr int QH = Q / 2;
        float* s_red = s_aux + 3 * cap + 4 * NB;
        const int px = x0 + wx + 4 * lk;
#pragma unroll
        for (int q = 0; q < QH; ++q) {
            const int co = q * 16 + li;
            const bool co_ok = co < p.cout;
            const float bias = (co_ok && p.bias) ? p.bias[co] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int y = y0 + wy + r;
                if (co_ok && y < p.h && px + 3 < p.w) {
                    f32x4 lo, hi;
                    lo[0] = acc[r][q][0] + bias; lo[1] = acc[r][q + QH][0] + bias; lo[2] = acc[r][q][1] + bias; lo[3] = acc[r][q + QH][1] + bias;
                    hi[0] = acc[r][q][2] + bias; hi[1] = acc[r][q + QH][2] + bias; hi[2] = acc[r][q][3] + bias; hi[3] = acc[r][q + QH][3] + bias;
                    float* dst = p.out + n * p.out_ns + static_cast<int64_t>(co) * p.out_cs + static_cast<int64_t>(2 * y + ph) * p.out_w + 2 * px;
                    *reinterpret_cast<f32x4*>(dst) = lo;
                    *reinterpret_cast<f32x4*>(dst + 4) = hi;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s1 += lo[e] + hi[e]; s2 += lo[e] * lo[e] + hi[e] * hi[e]; }
                }
            }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lk == 0) {
                s_red[((tid >> 6) * NB + q * 16 + li) * 2] = s1;
                s_red[((tid >> 6) * NB + q * 16 + li) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (p.out_sums && tid < 2 * 16 * QH) {          // no statistics in inference mode
            const int j = tid >> 1, which = tid & 1;
            if (j < p.cout) {
                double t = 0.0;
                for (int wv = 0; wv < 4; ++wv) t += static_cast<double>(s_red[(wv * NB + j) * 2 + which]);
                atomicAdd(p.out_sums + 2 * j + which, t);
            }
        }
        return;
    }
    conv_epilogue<Q, EPI, R>(po, acc, s_aux + 3 * cap, s_aux + 3 * cap + 4 * NB, x0, y0, wx, wy, co_base, n);
}

template <int KS, int KC, int Q, int IN, int EPI, int WX, int R, int NBUF, int MINW, int VEC, int XF = 0, int EXP = 0, int PH = -1, int BF = 0>
inline int launch_conv_dma_vec(ConvParams p, hipStream_t stream) {
    static_assert(XF == 0 || NBUF == 2 || IN != IN_BNRELU, "the in-place transform pipeline is written for two buffers");
    using G = ConvGeom<KS, KC, WX, R, VEC>;
    using S = ConvDmaSmem<KS, KC, Q, WX, R, NBUF, VEC, IN>;
    p.tiles_x = (p.w + G::kTileX - 1) / G::kTileX;
    p.bn_cap = (IN == IN_BNRELU) ? ((p.cin + KC - 1) / KC * KC + 15) / 16 * 16 : 0;
    const int tiles_y = (p.h + G::kTileY - 1) / G::kTileY;
    dim3 grid(p.tiles_x * tiles_y, p.ksplit > 0 ? p.ksplit : (PH >= 0 ? (PH == 2 ? 2 : 1) : (p.cout + 16 * Q - 1) / (16 * Q)), p.n);
    const size_t smem = S::bytes(p.bn_cap);
    static size_t configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& configured = configured_by_device[dev & 15];
    if (smem > 48 * 1024 && smem > configured) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dma_kernel<KS, KC, Q, IN, EPI, WX, R, NBUF, MINW, VEC, XF, EXP, PH, BF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
        configured = smem;
    }
    conv_dma_kernel<KS, KC, Q, IN, EPI, WX, R, NBUF, MINW, VEC, XF, EXP, PH, BF><<<grid, kConvThreads, smem, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// 16-byte DMA whenever the input rows are float4-aligned, dword DMA otherwise
template <int KS, int KC, int Q, int IN, int EPI, int WX, int R, int NBUF, int MINW = 1, int BF = 0>
inline int launch_conv_dma(const ConvParams& p, hipStream_t stream) {
    if constexpr (IN != IN_UPSAMPLE && IN != IN_UNPOOL) {
        const bool aligned = (p.w % 4 == 0) && (p.in_w % 4 == 0) && (p.in_cs % 4 == 0) && (p.in_ns % 4 == 0) &&
                             (reinterpret_cast<uintptr_t>(p.in) % 16 == 0);
        if (aligned) return launch_conv_dma_vec<KS, KC, Q, IN, EPI, WX, R, NBUF, MINW, 4, 0, 0, -1, BF>(p, stream);
    }
    return launch_conv_dma_vec<KS, KC, Q, IN, EPI, WX, R, NBUF, MINW, 1, 0, 0, -1, BF>(p, stream);
}

// tile-shape choice as launch_conv_auto; KC is the per-chunk depth of the double-buffered pipeline
template <int KS, int KC, int Q, int IN, int EPI, int BIG_R = 8, int NBUF = 2, int BIG_MINW = 1, int BF = 0>
inline int launch_conv_dma_auto(const ConvParams& p, hipStream_t stream) {
    const long tiles_big = static_cast<long>((p.w + 31) / 32) * ((p.h + 15) / 16) * p.n;
    if (tiles_big >= 512) return launch_conv_dma<KS, KC, Q, IN, EPI, 2, BIG_R, NBUF, BIG_MINW, BF>(p, stream);
    const long tiles_mid = static_cast<long>((p.w + 15) / 16) * ((p.h + 15) / 16) * p.n;
    if (tiles_mid >= 384) return launch_conv_dma<KS, KC, Q, IN, EPI, 1, 4, NBUF, 1, BF>(p, stream);
    constexpr int KCS = (NBUF == 2 && KS == 3 && KC <= 8) ? 16 : KC;
    return launch_conv_dma<KS, KCS, Q, IN, EPI, 1, 2, NBUF, 1, BF>(p, stream);
}

}  // namespace endo
