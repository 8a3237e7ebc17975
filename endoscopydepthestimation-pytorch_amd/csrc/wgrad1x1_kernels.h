// Weight gradient of the transition-down 1x1 convolution (reference models.py:56-67: BN -> ReLU -> conv1x1 ->
// [dropout] -> maxpool2) on the fp32 matrix cores, LDS-DMA staged:
//
//   dW[co][ci] += sum over full-resolution pixels  Gfull[co][p] * relu(bn(x[ci][p]))
//
// where Gfull is the max-pool un-routing of the pooled gradient G (non-zero only at the argmax position of each
// 2x2 window).  Neither Gfull nor the activation is materialised: a chunk is one pooled row segment = 2 rows x 32
// pixels, staged as raw x rows, pooled G rows and their argmax codes (one byte per pooled pixel); un-routing (byte
// compare, select) and BN + ReLU happen on the fragment read.
// GEMM view: M = 96 cout, N = 96 cin per block (3 x 3 MFMA tiles per wave), K = pixels; a block walks a strided
// subset of the chunks with two LDS buffers (one barrier per chunk) and ends with one fp32 atomic per dW element.
// (A 512-thread variant whose halves split the k-steps of a chunk measured 1.5 % slower in the training step.)
//
// Round 6: everything is staged by 16-BYTE DMA instructions (a wave issues one in ~42 cycles whatever its size --
// profiles/r06_dma_rate_probe.txt -- and the dword form needed 72 per wave and chunk against 4 600 cycles of MFMAs; now 8).
// An instruction's 64 lanes write 64 consecutive 16-byte units of LDS, so the image is made conflict-free on the SOURCE
// side: unit u of the x image holds channel c = u / 16, 4-pixel piece j = (u % 16) ^ (c % 8) (j = 8 row + piece of the row);
// unit u of the G image holds cout co = u / 4, 4-pooled-pixel piece k = (u % 4) ^ ((co / 2) % 4); the codes are one unit per
// cout.  A lane reads its operands of FOUR k-steps with one ds_read_b128 (x) and one ds_read_b64 (G): MFMA k-step (U, e), lane
// (li, lk) = pixel e of piece 4 U + lk -- the pixel order of the sum is free as long as both operands agree.
#pragma once

#include "conv_dma_kernels.h"
#include "wgrad_kernels.h"

namespace endo {

constexpr int kP1Seg = 32;                       // pixels per row of a chunk
// Block shape: QO x QI waves, each with a 48 cout x 48 cin quadrant (3 x 3 MFMA tiles).  2 x 2 (a 96 x 96 tile) where the channel counts are
// multiples of 96; 3 x 1 (144 couts x 48 cins) for the 144- and 240-channel transitions, which 96 x 96 tiles cover with 1.78 / 1.44
// times the MFMAs.
template <int QO, int QI>
struct P1Geom {
    static constexpr int kWaves = QO * QI;
    static constexpr int kTileO = 48 * QO, kTileI = 48 * QI;
    static constexpr int kXUnits = kTileI * 16;          // 16-byte units of the x image: 2 rows x 8 pieces per channel
    static constexpr int kGUnits = kTileO * 4;           // pooled gradients: 16 pooled pixels per cout
    static constexpr int kCUnits = kTileO;               // codes: 16 bytes per cout
    static constexpr int kXI = kXUnits / 64, kGI = kGUnits / 64, kCI = (kCUnits + 63) / 64;          // DMA instructions per chunk
    static constexpr int kBuf = 4 * (kXUnits + kGUnits + kCUnits);     // floats per buffer (2 x 2: 32 256 bytes)
    static constexpr size_t kBytes = 2 * kBuf * sizeof(float);
};

// BF: 1 = bf16 MFMA operands (ENDO_OPT_MFMA_BF16): the four pixels of a lane's piece form the k = 4 lk + e of one
// v_mfma_f32_16x16x16_bf16; un-routing and BN + ReLU stay fp32
template <int BF = 0, int QO = 2, int QI = 2>
__global__ void __launch_bounds__(64 * QO * QI, 2) wgrad1x1_dma_kernel(const WgradParams p) {
    using G = P1Geom<QO, QI>;
    constexpr int kP1XUnits = G::kXUnits, kP1GUnits = G::kGUnits, kP1Buf = G::kBuf;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int co_base = blockIdx.y * G::kTileO;
    const int ci_base = blockIdx.z * G::kTileI;
    const int segs = (p.w + kP1Seg - 1) / kP1Seg;
    const int chunks_per_sample = segs * (p.h >> 1);
    const int chunks_total = chunks_per_sample * p.n;
    const int wr = wave / QI, wc = wave % QI;      // wave's 48 x 48 quadrant of the block's tile

    float sc[3], sh[3];          // relu(sc * x + sh)
    int cur_grp = -1;
    auto load_consts = [&](int g) {           // BN constants of this lane's 3 input channels for sample group g
        const float* saved = p.saved + g * p.gs;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ch = ci_base + wc * 48 + j * 16 + li;
            sc[j] = 0.f; sh[j] = 0.f;          // channels past cin read zeros through the descriptor: relu(0 * 0 + 0) = 0
            if (ch < p.cin) {
                sc[j] = p.gamma[ch] * saved[2 * ch + 1];
                sh[j] = fmaf(-saved[2 * ch], sc[j], p.beta[ch]);
            }
        }
        cur_grp = g;
    };

    f32x4 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- DMA sources through buffer descriptors ----
    // A DMA instruction's address is descriptor base (the sample's planes) + a per-lane byte offset that never changes + ONE scalar byte
    // offset (chunk position + channel plane).  Channels past cin / cout lie past the descriptor's range and read zeros (the BN constants
    // of such channels are zero too), a piece right of the image gets a per-lane offset past the range: zeros on the gradient side, which
    // is what makes its products vanish.
    const unsigned kOob = 0x80000000u;
    // x: instruction T = channels 4 T .. + 3 of the tile; lane = (channel lane / 16, unit lane % 16)
    int x_piece[2];              // the piece this lane fetches in even / odd instructions: (lane % 16) ^ (channel % 8)
    unsigned x_lane[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int cl = 4 * o + (lane >> 4);
        x_piece[o] = (lane & 15) ^ (cl & 7);
        x_lane[o] = 4u * static_cast<unsigned>((lane >> 4) * p.in_cs + (x_piece[o] >> 3) * p.in_w + 4 * (x_piece[o] & 7));
    }
    // G: instruction T = couts 16 T .. + 15; lane = (cout lane / 4, unit lane % 4)
    const int g_piece = (lane & 3) ^ ((lane >> 3) & 3);
    const unsigned g_lane = 4u * static_cast<unsigned>((lane >> 2) * p.dy_cs + 4 * g_piece);
    // codes: one unit per cout, 64 couts per instruction
    const unsigned c_lane = static_cast<unsigned>(lane * p.dy_cs);
    const unsigned x_cs4 = 16u * static_cast<unsigned>(p.in_cs), dy_cs16 = 64u * static_cast<unsigned>(p.dy_cs);

    // the chunk walk (sample, pooled row, segment) advances by gridDim.x chunks per iteration: carried digit by digit, no division in the loop
    const int rows2 = p.h >> 1;
    int c_n, c_y2, c_seg;
    {
        const int c0 = blockIdx.x;
        c_n = c0 / chunks_per_sample;
        const int rem = c0 - c_n * chunks_per_sample;
        c_y2 = rem / segs;
        c_seg = rem - c_y2 * segs;
    }
    const int st = static_cast<int>(gridDim.x);
    const int st_n = st / chunks_per_sample, st_rem = st - st_n * chunks_per_sample;
    const int st_y2 = st_rem / segs, st_seg = st_rem - st_y2 * segs;
    auto advance = [&]() {
        c_seg += st_seg; c_y2 += st_y2; c_n += st_n;
        if (c_seg >= segs) { c_seg -= segs; ++c_y2; }
        if (c_y2 >= rows2) { c_y2 -= rows2; ++c_n; }
    };

    // issue the DMAs of the chunk the walk stands on into buffer `buf`
    auto issue = [&](int buf) {
        const int n = c_n, y2 = c_y2, xs = c_seg * kP1Seg;
        const WgSample sm(p, n);
        float* s_x = smem + buf * kP1Buf;
        float* s_g = s_x + 4 * kP1XUnits;
        float* s_c = s_g + 4 * kP1GUnits;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in + sm.in_off(p)), 0, p.cin * p.in_cs * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + sm.dy_off(p)), 0, p.cout * p.dy_cs * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.dy_idx + sm.idx_off(p)), 0, p.cout * p.dy_cs, 0x00020000);
        const unsigned x_vo0 = xs + 4 * (x_piece[0] & 7) < p.w ? x_lane[0] : kOob;
        const unsigned x_vo1 = xs + 4 * (x_piece[1] & 7) < p.w ? x_lane[1] : kOob;
        const unsigned x_so = static_cast<unsigned>(ci_base) * (x_cs4 >> 2) + 4u * static_cast<unsigned>(2 * y2 * p.in_w + xs);
        const int pxs = xs >> 1;
        const unsigned ppos = static_cast<unsigned>(y2 * p.dy_w + pxs);
        const unsigned g_vo = pxs + 4 * g_piece < (p.w >> 1) ? g_lane : kOob;
        // the chunk's kXI + kGI + kCI instructions dealt out to the waves in turn
        constexpr int kAll = G::kXI + G::kGI + G::kCI;
#pragma unroll
        for (int m = 0; m * G::kWaves < kAll; ++m) {
            const int L = m * G::kWaves + wave;
            if (L < G::kXI) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr_t)(s_x + L * 256), 16, (L & 1) ? x_vo1 : x_vo0, x_so + static_cast<unsigned>(L) * x_cs4, 0, 0);
            } else if (L < G::kXI + G::kGI) {
                const int T = L - G::kXI;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lptr_t)(s_g + T * 256), 16, g_vo,
                                                         static_cast<unsigned>(co_base + 16 * T) * (dy_cs16 >> 4) + 4u * ppos, 0, 0);
            } else if (L < kAll) {
                const int T = L - G::kXI - G::kGI;
                if (64 * T + lane < G::kCUnits)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(cr, (lptr_t)(s_c + T * 256), 16, c_lane,
                                                             static_cast<unsigned>(co_base + 64 * T) * static_cast<unsigned>(p.dy_cs) + ppos, 0, 0);
            }
        }
    };

    // operand addresses of this lane (dwords inside a buffer)
    //   x: channel c = wc * 48 + 16 j + li, piece 4 U + lk -> unit c * 16 + ((4 U + lk) ^ (li % 8))
    //   G: cout co = wr * 48 + 16 i + li, pooled piece k = 2 (U % 2) + lk / 2 -> unit co * 4 + (k ^ ((li / 2) % 4)), its half lk % 2
    int x_rd[4], g_rd[2];
#pragma unroll
    for (int U = 0; U < 4; ++U) x_rd[U] = 4 * ((wc * 48 + li) * 16 + ((4 * U + lk) ^ (li & 7)));
#pragma unroll
    for (int h = 0; h < 2; ++h) g_rd[h] = 4 * kP1XUnits + 4 * ((wr * 48 + li) * 4 + ((2 * h + (lk >> 1)) ^ ((li >> 1) & 3))) + 2 * (lk & 1);
    const int c_rd = 4 * (kP1XUnits + kP1GUnits) + 4 * (wr * 48 + li);
    const unsigned code_shift = 16u * (lk & 1);

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto compute = [&](int buf) {
        const float* s = smem + buf * kP1Buf;
        // the two code bytes of this lane's piece, for both halves of the row (pieces 0 .. 3 / 4 .. 7): dword (piece / 2) of the cout's unit, half piece % 2
        unsigned cp[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const u32x4_bits cw = *reinterpret_cast<const u32x4_bits*>(s + c_rd + i * 64);
#pragma unroll
            for (int h = 0; h < 2; ++h) cp[i][h] = ((lk >> 1) ? cw[2 * h + 1] : cw[2 * h]) >> code_shift;
        }
#pragma unroll
        for (int U = 0; U < 4; ++U) {
            // U = 2 row + half: this lane's piece is 4 (U % 2) + lk of image row U / 2
            f32x4 xv[3];
            f32x2 gv[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) xv[j] = *reinterpret_cast<const f32x4*>(s + x_rd[U] + j * 16 * 64);
#pragma unroll
            for (int i = 0; i < 3; ++i) gv[i] = *reinterpret_cast<const f32x2*>(s + g_rd[U & 1] + i * 16 * 16);
            float a[4][3], b[4][3];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned want = 2u * (U >> 1) + (e & 1);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const unsigned code = (cp[i][U & 1] >> (8 * (e >> 1))) & 0xffu;
                    a[e][i] = code == want ? gv[i][e >> 1] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) b[e][j] = __builtin_fmaxf(fmaf(xv[j][e], sc[j], sh[j]), 0.f);
            }
            if constexpr (BF != 0) {
                bf16x4_bits ap[3], bp[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    ap[i] = pack_bf16x4(a[0][i], a[1][i], a[2][i], a[3][i]);
                    bp[i] = pack_bf16x4(b[0][i], b[1][i], b[2][i], b[3][i]);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ap[i], bp[j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e][i], b[e][j], acc[i][j], 0, 0, 0);
            }
        }
    };

    int chunk = blockIdx.x;
    if (chunk < chunks_total) issue(0);
    int b = 0;
    for (; chunk < chunks_total; chunk += gridDim.x, b ^= 1) {
        const int g = WgSample(p, c_n).grp;          // the group of the chunk about to be computed (the walk still stands on it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        advance();
        if (chunk + static_cast<int>(gridDim.x) < chunks_total) issue(b ^ 1);
        if (g != cur_grp) load_consts(g);
        compute(b);
    }
    // lane holds D[co = 4*lk + e][ci = li] of each 16 x 16 sub-tile
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = co_base + wr * 48 + i * 16 + 4 * lk + e;
                const int ci = ci_base + wc * 48 + j * 16 + li;
                if (p.dw && co < p.cout && ci < p.cin) atomicAdd(p.dw + static_cast<int64_t>(co) * p.cin + ci, acc[i][j][e]);          // (p.dw == nullptr: tools/tdw_bench's build without the final atomics)
            }
}

// needs whole code dwords per pooled row segment (pooled width % 4 == 0) and even H, W
inline bool wgrad1x1_dma_ok(const WgradParams& p) {
    // (channel planes are addressed through buffer descriptors with 32-bit byte ranges and offsets: a tile past the last channel must still fit)
    const bool fits = (static_cast<int64_t>(p.cin) + 96) * p.in_cs * 4 < (1ll << 31) && (static_cast<int64_t>(p.cout) + 240) * p.dy_cs * 4 < (1ll << 31);
    // 16-byte pieces: 4 pixels of an x row, 4 pooled pixels of a gradient row (their addresses 16-byte aligned), 16 codes (4-byte aligned)
    return fits && (p.dy_w % 4 == 0) && (p.dy_cs % 4 == 0) && (p.dy_ns % 4 == 0) && (p.idx_ns % 4 == 0) && (p.h % 2 == 0) && (p.w == 2 * p.dy_w) &&
           (p.in_w % 4 == 0) && (p.in_cs % 4 == 0) && (p.in_ns % 4 == 0) && (p.gs % 4 == 0) && (p.in_gs % 4 == 0) &&
           (reinterpret_cast<uintptr_t>(p.in) % 16 == 0) && (reinterpret_cast<uintptr_t>(p.dy) % 16 == 0) && (reinterpret_cast<uintptr_t>(p.dy_idx) % 4 == 0);
}

template <int BF, int QO, int QI>
inline int launch_wgrad1x1_dma_shape(const WgradParams& p, hipStream_t stream) {
    using G = P1Geom<QO, QI>;
    const int tiles_co = (p.cout + G::kTileO - 1) / G::kTileO;
    const int tiles_ci = (p.cin + G::kTileI - 1) / G::kTileI;
    const int chunks_total = ((p.w + kP1Seg - 1) / kP1Seg) * (p.h / 2) * p.n;
    const int per_cu = std::min(static_cast<int>(160 * 1024 / G::kBytes), 12 / G::kWaves);          // blocks per CU: LDS, three waves per SIMD
    int splits = 256 * per_cu / (tiles_co * tiles_ci);
    if (splits < 1) splits = 1;
    if (splits > chunks_total) splits = chunks_total;
    static bool configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& configured = configured_by_device[dev & 15];
    if (!configured) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad1x1_dma_kernel<BF, QO, QI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(G::kBytes)));
        configured = true;
    }
    wgrad1x1_dma_kernel<BF, QO, QI><<<dim3(splits, tiles_co, tiles_ci), 64 * G::kWaves, G::kBytes, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// shape: 0 = the one whose tiles cover cout x cin with the fewer MFMAs (2 x 2 on a tie), 1 / 2 = 2 x 2, 3 x 1 (tools/tdw_bench; 5 x 1 waves on
// 240 x 48 tiles measured 89 us against 68 at 240 channels, level 3)
template <int BF = 0>
inline int launch_wgrad1x1_dma(const WgradParams& p, hipStream_t stream, int shape = 0) {
    if (shape == 0) {
        auto padded = [&](int to, int ti) { return static_cast<long>((p.cout + to - 1) / to) * to * ((p.cin + ti - 1) / ti) * ti; };
        shape = padded(96, 96) <= padded(144, 48) ? 1 : 2;
    }
    return shape == 1 ? launch_wgrad1x1_dma_shape<BF, 2, 2>(p, stream) : launch_wgrad1x1_dma_shape<BF, 3, 1>(p, stream);
}

}  // namespace endo
