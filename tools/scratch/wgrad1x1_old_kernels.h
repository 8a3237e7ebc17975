// Weight gradient of the transition-down 1x1 convolution (reference models.py:56-67: BN -> ReLU -> conv1x1 ->
// [dropout] -> maxpool2) on the fp32 matrix cores, LDS-DMA staged:
//
//   dW[co][ci] += sum over full-resolution pixels  Gfull[co][p] * relu(bn(x[ci][p]))
//
// where Gfull is the max-pool un-routing of the pooled gradient G (non-zero only at the argmax position of each
// 2x2 window).  Neither Gfull nor the activation is materialised: a chunk is one pooled row segment = 2 rows x 32
// pixels, staged as raw x rows (64 dwords each), pooled G rows (16 dwords) and their argmax codes (4 dwords, one
// byte per pooled pixel); un-routing (byte extract, compare, select) and BN+ReLU happen on the fragment read.
// GEMM view: M = 96 cout, N = 96 cin per block (3 x 3 MFMA tiles per wave), K = pixels; a block walks a strided
// subset of the chunks with two LDS buffers (one barrier per chunk) and ends with one fp32 atomic per dW element.
// (A 512-thread variant whose halves split the k-steps of a chunk measured 1.5 % slower in the training step.)
// LDS rows: x stride 66 (== 2 mod 32) and G stride 21 (odd): both fragment reads are bank-conflict free.


#include "../../endoscopydepthestimation-pytorch_amd/csrc/conv_dma_kernels.h"
#include "../../endoscopydepthestimation-pytorch_amd/csrc/wgrad_kernels.h"

namespace endo {

constexpr int kO1Tile = 96;
constexpr int kO1Seg = 32;                       // pixels per row of a chunk
constexpr int kO1ActStride = 2 * kO1Seg + 2;
constexpr int kO1DyStride = kO1Seg / 2 + kO1Seg / 8 + 1;
constexpr int kO1Buf = kO1Tile * (kO1ActStride + kO1DyStride);     // floats per buffer
constexpr size_t kO1Bytes = 2 * kO1Buf * sizeof(float);

// BF: 1 = bf16 MFMA operands (ENDO_OPT_MFMA_BF16): four consecutive k-steps (pixels 4 ks + lk, ks = 4 g .. 4 g + 3) form the
// k = 4 lk + i of one v_mfma_f32_16x16x16_bf16; un-routing and BN + ReLU stay fp32
template <int BF = 0>
__global__ void __launch_bounds__(kConvThreads) wgrad1x1_dma_old_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int co_base = blockIdx.y * kO1Tile;
    const int ci_base = blockIdx.z * kO1Tile;
    const int segs = (p.w + kO1Seg - 1) / kO1Seg;
    const int chunks_per_sample = segs * (p.h >> 1);
    const int chunks_total = chunks_per_sample * p.n;
    const int wr = wave >> 1, wc = wave & 1;      // wave's 48 x 48 quadrant of the 96 x 96 tile

    float sc[3], mn[3], bt[3];
    int cur_grp = -1;
    auto load_consts = [&](int g) {           // BN constants of this lane's 3 input channels for sample group g
        const float* saved = p.saved + g * p.gs;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int ch = ci_base + wc * 48 + j * 16 + li;
            sc[j] = 0.f; mn[j] = 0.f; bt[j] = 0.f;       // rows past cin hold the NaN pad: max(fma(NaN, 0, 0), 0) = 0
            if (ch < p.cin) {
                mn[j] = saved[2 * ch];
                sc[j] = p.gamma[ch] * saved[2 * ch + 1];
                bt[j] = p.beta[ch];
            }
        }
        cur_grp = g;
    };

    f32x4 acc[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- DMA sources through buffer descriptors (round 5) ----
    // A DMA instruction's address is descriptor base (the sample's planes) + a per-lane byte offset that never changes + ONE scalar byte
    // offset (chunk position + channel plane), advanced by a constant per instruction: one scalar add per DMA instead of the 64-bit
    // multiply-add, range compare and pad-pointer select of the pointer form -- its scalar instructions were 18 % of the wave's issue time.
    // Channels past cin / cout lie past the descriptor's range and read zeros (the BN constants of such rows are zero too, load_consts),
    // a pixel right of the image gets a per-lane offset past the range: zeros on the gradient side, which is what makes its product vanish.
    const int arow = lane >> 5, ax = lane & 31;
    const unsigned kOob = 0x80000000u;
    const unsigned x_lane = 4u * static_cast<unsigned>(arow * p.in_w + ax);
    const bool is_dy = lane < kO1Seg / 2, is_code = lane >= kO1Seg / 2 && lane < kO1Seg / 2 + kO1Seg / 8;
    const int dcol = is_dy ? lane : 4 * (lane - kO1Seg / 2);              // pooled column inside the segment
    const unsigned d_lane = is_dy ? 4u * static_cast<unsigned>(lane) : 4u * static_cast<unsigned>(lane - kO1Seg / 2);
    const unsigned x_cs = 4u * static_cast<unsigned>(p.in_cs), dy_csb = 4u * static_cast<unsigned>(p.dy_cs), code_csb = static_cast<unsigned>(p.dy_cs);
    const unsigned x_ch0 = static_cast<unsigned>(ci_base + wave * (kO1Tile / 4)) * x_cs;
    const unsigned dy_ch0 = static_cast<unsigned>(co_base + wave * (kO1Tile / 4)) * dy_csb;
    const unsigned code_ch0 = static_cast<unsigned>(co_base + wave * (kO1Tile / 4)) * code_csb;

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
        const int n = c_n, y2 = c_y2, xs = c_seg * kO1Seg;
        const WgSample sm(p, n);
        float* s_act = smem + buf * kO1Buf;
        float* s_dy = s_act + kO1Tile * kO1ActStride;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in + sm.in_off(p)), 0, p.cin * p.in_cs * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + sm.dy_off(p)), 0, p.cout * p.dy_cs * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.dy_idx + sm.idx_off(p)), 0, p.cout * p.dy_cs, 0x00020000);
        const unsigned x_vo = xs + ax < p.w ? x_lane : kOob;
        unsigned so = x_ch0 + 4u * static_cast<unsigned>(2 * y2 * p.in_w + xs);
#pragma unroll
        for (int t = 0; t < kO1Tile / 4; ++t) {
            const int r = wave * (kO1Tile / 4) + t;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr_t)(s_act + r * kO1ActStride), 4, x_vo, so, 0, 0);
            so += x_cs;
        }
        const int pxs = xs >> 1;
        const unsigned d_vo = pxs + dcol < (p.w >> 1) ? d_lane : kOob;
        const unsigned ppos = static_cast<unsigned>(y2 * p.dy_w + pxs);
        if (is_dy) {          // lanes 0 .. 15: the pooled gradient row segment
            unsigned sg = dy_ch0 + 4u * ppos;
#pragma unroll
            for (int t = 0; t < kO1Tile / 4; ++t) {
                const int r = wave * (kO1Tile / 4) + t;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lptr_t)(s_dy + r * kO1DyStride), 4, d_vo, sg, 0, 0);
                sg += dy_csb;
            }
        }
        if (is_code) {        // lanes 16 .. 19: its argmax codes, four pooled pixels per dword; they land behind the 16 gradient dwords
            unsigned sc2 = code_ch0 + ppos;
#pragma unroll
            for (int t = 0; t < kO1Tile / 4; ++t) {
                const int r = wave * (kO1Tile / 4) + t;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(cr, (lptr_t)(s_dy + r * kO1DyStride), 4, d_vo, sc2, 0, 0);
                sc2 += code_csb;
            }
        }
    };

    const unsigned lane_shift = 8u * (lk >> 1);
    const unsigned lane_want = lk & 1;
    auto compute = [&](int buf) {
        const float* s_act = smem + buf * kO1Buf;
        const float* s_dy = s_act + kO1Tile * kO1ActStride;
        const float* g_base = s_dy + (wr * 48 + li) * kO1DyStride;
        const float* b_base = s_act + (wc * 48 + li) * kO1ActStride + lk;
        unsigned cw[3][4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int d = 0; d < 4; ++d) cw[i][d] = __float_as_uint(g_base[i * 16 * kO1DyStride + kO1Seg / 2 + d]);
        auto operands = [&](int ks, float (&a)[3], float (&b)[3]) {
            // k = pixel 4*ks + lk of the chunk: row ks>>3, x = 4*(ks&7) + lk; pooled column x>>1, code 2*row + (x&1)
            const int pc = 2 * (ks & 7) + (lk >> 1);
            const unsigned want = 2u * (ks >> 3) + lane_want;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float g = g_base[i * 16 * kO1DyStride + pc];
                const unsigned code = (cw[i][(ks & 7) >> 1] >> (16u * (ks & 1) + lane_shift)) & 0xffu;
                a[i] = code == want ? g : 0.f;
                const float v = b_base[i * 16 * kO1ActStride + 4 * ks];
                b[i] = __builtin_fmaxf(fmaf(v - mn[i], sc[i], bt[i]), 0.f);
            }
        };
        if constexpr (BF != 0) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float a[4][3], b[4][3];
#pragma unroll
                for (int i = 0; i < 4; ++i) operands(4 * g4 + i, a[i], b[i]);
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
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                float a[3], b[3];
                operands(ks, a, b);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
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
                if (co < p.cout && ci < p.cin) atomicAdd(p.dw + static_cast<int64_t>(co) * p.cin + ci, acc[i][j][e]);
            }
}

// needs whole code dwords per pooled row segment (pooled width % 4 == 0) and even H, W
inline bool wgrad1x1_dma_old_ok(const WgradParams& p) {
    // (channel planes are addressed through buffer descriptors with 32-bit byte ranges and offsets: a tile past the last channel must still fit)
    const bool fits = (static_cast<int64_t>(p.cin) + kO1Tile) * p.in_cs * 4 < (1ll << 31) && (static_cast<int64_t>(p.cout) + kO1Tile) * p.dy_cs * 4 < (1ll << 31);
    return fits && (p.dy_w % 4 == 0) && (p.dy_cs % 4 == 0) && (p.idx_ns % 4 == 0) && (p.h % 2 == 0) && (p.w % 2 == 0) &&
           (reinterpret_cast<uintptr_t>(p.dy_idx) % 4 == 0);
}

template <int BF = 0>
inline int launch_wgrad1x1_dma_old(const WgradParams& p, hipStream_t stream) {
    const int tiles_co = (p.cout + kO1Tile - 1) / kO1Tile;
    const int tiles_ci = (p.cin + kO1Tile - 1) / kO1Tile;
    const int chunks_total = ((p.w + kO1Seg - 1) / kO1Seg) * (p.h / 2) * p.n;
    int splits = 512 / (tiles_co * tiles_ci);          // 2 blocks per CU (LDS)
    if (splits < 1) splits = 1;
    if (splits > chunks_total) splits = chunks_total;
    static bool configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& configured = configured_by_device[dev & 15];
    if (!configured) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad1x1_dma_old_kernel<BF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(kO1Bytes)));
        configured = true;
    }
    wgrad1x1_dma_old_kernel<BF><<<dim3(splits, tiles_co, tiles_ci), kConvThreads, kO1Bytes, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

}  // namespace endo
