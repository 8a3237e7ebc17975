// Dense-layer forward (BN -> ReLU -> conv3x3, growth 12; reference models.py:19-28) in Winograd F(2x2, 3x3) form on the
// fp32 matrix cores: 16 instead of 36 multiply-accumulates per 2x2 output tile, input channel and output channel.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A          d_c: 4x4 patch of relu(bn(x_c)), g_c: 3x3 filter, Y: 2x2 outputs
//
// The 16 element-wise products are 16 independent GEMMs over the input channels, and that is how they run:
//   MFMA roles (v_mfma_f32_16x16x4_f32), one per transform-domain position xi = 0..15:
//     A[i = tile][k = channel] = V_xi = (B^T d B)[xi]   computed by the lane that owns (tile i, channel k) from ITS 4x4 patch
//     B[k = channel][j = cout] = U_xi = (G g G^T)[xi]   pre-transformed once per forward pass (wino_fwd_weights_kernel)
//     D_xi[i][j] accumulates over the whole K loop; the output transform A^T M A is per lane, in registers.
// An M-group is a row of 16 tiles (32 x 2 output pixels); a wave owns R of them, a block 4 R (32 x 8R pixels).
// Everything around the arithmetic is the LDS-DMA pipeline of conv_dma_kernels.h: K-chunks of KC input channels (haloed
// tile + the chunk's U slice) by 16-byte global_load_lds into two buffers, one barrier per chunk, BN+ReLU applied on the
// fragment read with the NaN pad standing in for the zero padding of the post-activation tensor, per-channel sum / sum^2
// of the stored values for the BN layers that follow.
//
// Rounding: the input transform only adds (|V| <= 4 max|d|), U carries the 1/2 and 1/4 of G; measured against fp64 on
// relu-like data with 180 input channels the result is as close as the direct fp32 accumulation (4e-7 vs 1e-6 of the
// output's maximum, tools note in DESIGN.md) -- the long K sum dominates either way.
#pragma once

#include "conv_dma_kernels.h"

namespace endo {

constexpr int kWinoUStride = 16 * 16 + 16;          // floats per input channel of U: [xi 16][j 16] + pad (== 16 mod 32: conflict-free B reads)
constexpr int kWinoMaxLayers = 48;

// ---- weights: U[ci][xi][j] = (G g G^T)[xi], g = W[j][ci][3][3]; columns j >= cout are zero -------------------------------------
struct WinoWeightTable {
    int layers;
    int start[kWinoMaxLayers + 1];          // prefix sum of cin * 16 work items
    int cin[kWinoMaxLayers];
    int cout[kWinoMaxLayers];
    int64_t w_off[kWinoMaxLayers];          // floats from the parameter base
    int64_t u_off[kWinoMaxLayers];          // floats from the U base
    // 0 = U as above; 1 = the layer runs the direct kernel on K-chunks of 16 channels this pass: its slot holds the ORIGINAL weights in that
    // pipeline's LDS order instead, [chunk][tap 9][channel 16][cout 16] with zeros for missing channels / outputs (ConvParams::wgt_chunks;
    // 144 floats per input channel fit the slot's 272)
    int mode[kWinoMaxLayers];
};
constexpr int kDenseChunkFloats = 9 * 16 * 16;

__global__ void __launch_bounds__(256) wino_fwd_weights_kernel(const WinoWeightTable t, const float* __restrict__ params, float* __restrict__ u) {
    const int total = t.start[t.layers];
    for (int item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += gridDim.x * blockDim.x) {
        int l = 0;
        while (item >= t.start[l + 1]) ++l;
        const int e = item - t.start[l];
        const int ci = e >> 4, j = e & 15;
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = 0.f;
        if (j < t.cout[l]) {
            const float* src = params + t.w_off[l] + (static_cast<int64_t>(j) * t.cin[l] + ci) * 9;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) g[a][b] = src[a * 3 + b];
        }
        if (t.mode[l] == 1) {          // the direct kernel's chunk order (the last channel's thread also zeroes the channels that pad the last chunk)
            float* dst = u + t.u_off[l] + static_cast<int64_t>(ci >> 4) * kDenseChunkFloats + (ci & 15) * 16 + j;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) dst[tap * 256] = g[tap / 3][tap % 3];
            if (ci == t.cin[l] - 1)
                for (int cz = (ci & 15) + 1; cz < 16; ++cz)
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) dst[tap * 256 + (cz - (ci & 15)) * 16] = 0.f;
            continue;
        }
        float h[4][3];          // G g
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            h[0][b] = g[0][b];
            h[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            h[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            h[3][b] = g[2][b];
        }
        float* dst = u + t.u_off[l] + static_cast<int64_t>(ci) * kWinoUStride + j;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            dst[(4 * a + 0) * 16] = h[a][0];
            dst[(4 * a + 1) * 16] = 0.5f * (h[a][0] + h[a][1] + h[a][2]);
            dst[(4 * a + 2) * 16] = 0.5f * (h[a][0] - h[a][1] + h[a][2]);
            dst[(4 * a + 3) * 16] = h[a][2];
        }
        if (j == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) dst[256 + k] = 0.f;          // the pad is DMA'd along with the rest
        }
    }
}

template <int R, int KC, int NS = 2>
struct WinoFwdGeom {
    static constexpr int kTileX = 32;
    static constexpr int kTileY = 8 * R;                      // 4 waves x R tile rows x 2 pixel rows
    static constexpr int kLeft = 4;                           // the tile starts 4 pixels left of the output tile: rows of whole aligned float4s
    static constexpr int kCols = kTileX + 2 * kLeft;          // 40
    static constexpr int kRows = kTileY + 2;
    static constexpr int kPlane = kRows * kCols;
    // channel stride == 32 (mod 64) dwords: the patch reads are three aligned 8-byte reads per row (below), whose 64-bank map puts the
    // two channels of a half-wave (lk, lk + 1) on disjoint halves -- conflict-free.  (Round 2 read columns 0 and 3 of the patch as
    // dwords at a stride of 2 with a channel stride == 16 (mod 32): the two channels met on the same 16 odd banks, a 2-way conflict
    // on every such read -- SQ_LDS_BANK_CONFLICT 32 % of SQ_LDS_IDX_ACTIVE in profiles/r02_c_sq_counters.txt.)
    static constexpr int kCS = ((kPlane - 32 + 63) / 64) * 64 + 32;
    static constexpr int kUnits = kCS / 4;                    // 16-byte DMA units per channel, the kCS - kPlane pad floats included
    static constexpr int kPos = (kUnits + kConvThreads - 1) / kConvThreads;
    static constexpr int kUUnits = KC * kWinoUStride / 4;
    static constexpr int kUPos = (kUUnits + kConvThreads - 1) / kConvThreads;
    // one flat list of the chunk's DMA units (input tile of KC channels, then the U slice) dealt out round-robin: every wave
    // issues kDma instructions per chunk, the last wave(s) one fewer when the final round does not reach them (kDmaMin), so
    // "all but the newest k chunks have landed" is guaranteed by s_waitcnt vmcnt(k * kDmaMin) (loads retire in order)
    static constexpr int kFlat = KC * kUnits + kUUnits;
    static constexpr int kDma = (kFlat + kConvThreads - 1) / kConvThreads;
    static constexpr int kDmaMin = (kFlat - (kDma - 1) * kConvThreads > 3 * 64) ? kDma : kDma - 1;
    static_assert(kCS == 4 * kUnits, "a stage is one contiguous array of DMA units");
    static constexpr int kBuf = KC * kCS + KC * kWinoUStride;          // floats per stage
    static constexpr int kTail = 4 * 16 * 2;                  // statistics scratch: [4 waves][16][2]
    static size_t bytes(int bn_cap) { return sizeof(float) * (NS * kBuf + 3 * bn_cap + kTail); }
};

// p.wgt = this layer's U (kWinoUStride floats per input channel), p.cout <= 16, p.w % 4 == 0, 16-byte aligned planes.
// NS: LDS stages.  The Winograd form does 4/9 of the arithmetic on the same input bytes, so a K-chunk's MFMAs (~0.5 us) no
// longer cover the latency of the next chunk's DMA (~1.5 us): with two stages the kernel ran at 42 % MFMA utilisation and
// 2 TB/s, bound by neither.  NS stages keep NS - 1 chunks in flight.
// EXP: diagnostic bit mask (0 in the product; timing only, results are wrong): 1 = only the first NS - 1 chunks are DMA'd,
// 2 = no BN + ReLU, 4 = no input transform, 8 = patch values are constants (no LDS reads), 16 = U values are constants,
// 32 = no chunk barrier / DMA wait, 64 = no MFMAs
template <int R, int KC, int MINW, int NS, int EXP = 0>
__global__ void __launch_bounds__(kConvThreads, MINW) wino_fwd_kernel(const ConvParams p0) {
    using G = WinoFwdGeom<R, KC, NS>;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_aux = smem + NS * G::kBuf;
    int grp, n;
    group_of(p0, blockIdx.z, grp, n);
    const ConvParams p = group_view(p0, grp);
    const int groups = p0.group_n > 0 ? gridDim.z / p0.group_n : 1;
    const bool first_of_group = blockIdx.x == 0 && n == 0;
    const int cap = p.bn_cap;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int tile = (gridDim.x & 7) == 0 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int x0 = (tile % p.tiles_x) * G::kTileX;
    const int y0 = (tile / p.tiles_x) * G::kTileY;

    for (int c = tid; c < p.cin; c += kConvThreads) {
        float scale, mean, beta;
        bn_input_constants(p, p0, grp, groups, first_of_group, c, scale, mean, beta);
        s_aux[c] = scale;
        s_aux[cap + c] = mean;
        s_aux[2 * cap + c] = beta;
    }
    for (int c = p.cin + tid; c < ((p.cin + KC - 1) / KC) * KC; c += kConvThreads) {
        s_aux[c] = 0.f; s_aux[cap + c] = 0.f; s_aux[2 * cap + c] = 0.f;          // their raw values are the NaN pad
    }

    f32x4 acc[R][16];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) acc[r][xi] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this thread's DMA units of a chunk (the same every chunk): a running source pointer per unit that advances by a fixed
    // byte stride per chunk (KC input planes, KC rows of U, or 0 for a padding unit) -- two VALU per DMA and no scalar
    // address arithmetic in the K loop.  p.cin % KC == 0 (wino_fwd_ok), so a chunk never needs a channel-range check.
    const float* d_ptr[G::kDma];
    unsigned d_stride[G::kDma];
    {
        const float* in_n = p.in + n * p.in_ns;
#pragma unroll
        for (int k = 0; k < G::kDma; ++k) {
            const int e = tid + k * kConvThreads;
            d_ptr[k] = g_pad_consts;          // NaN pad: out-of-image pixels of a BN+ReLU input
            d_stride[k] = 0;
            if (e < KC * G::kUnits) {
                const int c = e / G::kUnits, u = e - c * G::kUnits;
                const int ry = u / (G::kCols / 4);
                const int rx = (u - ry * (G::kCols / 4)) * 4;
                const int gy = y0 - 1 + ry;
                const int gx = x0 - G::kLeft + rx;
                if (ry < G::kRows && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) {
                    d_ptr[k] = in_n + static_cast<int64_t>(c) * p.in_cs + gy * p.in_w + gx;
                    d_stride[k] = static_cast<unsigned>(KC) * static_cast<unsigned>(p.in_cs) * 4u;
                }
            } else if (e < G::kFlat) {
                d_ptr[k] = p.wgt + 4 * (e - KC * G::kUnits);
                d_stride[k] = KC * kWinoUStride * 4u;
            }
        }
    }
    const int nchunks = p.cin / KC;

    // chunks are issued in order, each exactly once: the pointers simply walk
    auto issue_dma = [&](int buf) {
        float* s_stage = smem + buf * G::kBuf + wave * 256;          // the wave's 64 units of round k land at 16 (256 k + 64 wave) bytes and up:
#pragma unroll                                                    // a stage is one contiguous array of units in list order (kCS == 4 kUnits)
        for (int k = 0; k < G::kDma; ++k) {
            if (k + 1 < G::kDma) {          // full rounds: every lane of every wave has a unit
                __builtin_amdgcn_global_load_lds((gptr_t)d_ptr[k], (lptr_t)(s_stage + k * 4 * kConvThreads), 16, 0, 0);
            } else {                        // last round: partially filled
                const int e0 = k * kConvThreads + wave * 64;
                if (e0 < G::kFlat && e0 + lane < G::kFlat)
                    __builtin_amdgcn_global_load_lds((gptr_t)d_ptr[k], (lptr_t)(s_stage + k * 4 * kConvThreads), 16, 0, 0);
            }
            d_ptr[k] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(d_ptr[k]) + d_stride[k]);
        }
    };

    auto compute = [&](int chunk, int buf) {
        const float* s_in = smem + buf * G::kBuf;
        const float* s_u = s_in + KC * G::kCS;
#pragma unroll
        for (int quad = 0; quad < KC / 4; ++quad) {
            const int ch = chunk * KC + quad * 4 + lk;
            const float sc = s_aux[ch], mn = s_aux[cap + ch], bt = s_aux[2 * cap + ch];
            const f32x2 mn2 = {mn, mn}, sc2 = {sc, sc}, bt2 = {bt, bt};
            const float* b_base = s_u + (quad * 4 + lk) * kWinoUStride + li;
            float b[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) b[xi] = (EXP & 16) ? static_cast<float>(xi + lane) : b_base[xi * 16];
            // the lane's patch rows: LDS rows 2 t0 .. 2 t0 + 2R + 1 of its R tile rows (consecutive tile rows share two rows),
            // LDS columns 2i+3 .. 2i+6 out of the three aligned pairs (2i+2, 2i+3), (2i+4, 2i+5), (2i+6, 2i+7): 8-byte reads only.
            // BN + ReLU once per value; max(NaN, 0) = 0 turns the NaN pad into the zero padding of the post-activation tensor.
            const float* a_base = s_in + (quad * 4 + lk) * G::kCS + (2 * wave * R) * G::kCols + 2 * li + 2;
            f32x2 mid[2 * R + 2], end[2 * R + 2];
#pragma unroll
            for (int row = 0; row < 2 * R + 2; ++row) {
                f32x2 m, e;
                if constexpr ((EXP & 8) != 0) {
                    m = f32x2{sc + row, mn + row}; e = f32x2{bt + row, sc - row};
                } else {
                    const f32x2 left = *reinterpret_cast<const f32x2*>(a_base + row * G::kCols);
                    const f32x2 right = *reinterpret_cast<const f32x2*>(a_base + row * G::kCols + 4);
                    m = *reinterpret_cast<const f32x2*>(a_base + row * G::kCols + 2);
                    e = f32x2{left[1], right[0]};
                }
                if constexpr ((EXP & 2) == 0) {
                    m = __builtin_elementwise_fma(m - mn2, sc2, bt2);
                    e = __builtin_elementwise_fma(e - mn2, sc2, bt2);
                    mid[row] = f32x2{__builtin_fmaxf(m[0], 0.f), __builtin_fmaxf(m[1], 0.f)};
                    end[row] = f32x2{__builtin_fmaxf(e[0], 0.f), __builtin_fmaxf(e[1], 0.f)};
                } else {
                    mid[row] = m; end[row] = e;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3); then (.) B: columns (c0 - c2, c1 + c2, c2 - c1, c1 - c3)
                f32x2 tm[4], te[4];
                if constexpr ((EXP & 4) == 0) {
                    tm[0] = mid[2 * r] - mid[2 * r + 2];     te[0] = end[2 * r] - end[2 * r + 2];
                    tm[1] = mid[2 * r + 1] + mid[2 * r + 2]; te[1] = end[2 * r + 1] + end[2 * r + 2];
                    tm[2] = mid[2 * r + 2] - mid[2 * r + 1]; te[2] = end[2 * r + 2] - end[2 * r + 1];
                    tm[3] = mid[2 * r + 1] - mid[2 * r + 3]; te[3] = end[2 * r + 1] - end[2 * r + 3];
                } else {
#pragma unroll
                    for (int a = 0; a < 4; ++a) { tm[a] = mid[2 * r + a]; te[a] = end[2 * r + a]; }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float c0 = te[a][0], c1 = tm[a][0], c2 = tm[a][1], c3 = te[a][1];
                    float v0 = c0 - c2, v1 = c1 + c2, v2 = c2 - c1, v3 = c1 - c3;
                    if constexpr ((EXP & 4) != 0) { v0 = c0; v1 = c1; v2 = c2; v3 = c3; }
                    if constexpr ((EXP & 64) != 0) {
                        acc[r][4 * a + 0][0] += v0 * b[4 * a]; acc[r][4 * a + 1][0] += v1 * b[4 * a + 1];
                        acc[r][4 * a + 2][0] += v2 * b[4 * a + 2]; acc[r][4 * a + 3][0] += v3 * b[4 * a + 3];
                    } else {
                        acc[r][4 * a + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, b[4 * a + 0], acc[r][4 * a + 0], 0, 0, 0);
                        acc[r][4 * a + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, b[4 * a + 1], acc[r][4 * a + 1], 0, 0, 0);
                        acc[r][4 * a + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v2, b[4 * a + 2], acc[r][4 * a + 2], 0, 0, 0);
                        acc[r][4 * a + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v3, b[4 * a + 3], acc[r][4 * a + 3], 0, 0, 0);
                    }
                }
            }
        }
    };

    // NS - 1 chunks in flight.  Every wave issues exactly kDma DMA instructions per chunk, so when chunk c is about to be
    // consumed the newer chunks c + 1 .. c + NS - 2 may still be outstanding: vmcnt((NS - 2) * kDmaMin) (fewer near the end).
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0)
        if (s0 < nchunks) issue_dma(s0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int b = chunk % NS;
        const int newer = min(nchunks - 1 - chunk, NS - 2);          // chunks issued after this one (block-uniform)
        if ((EXP & 32) && chunk > 0) {}
        else if (newer >= NS - 2 && NS > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G::kDmaMin) : "memory");
        else if (newer == 1 && NS > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::kDmaMin) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(EXP & 32) || chunk == 0) __syncthreads();           // everybody's chunk has landed; stage (chunk - 1) % NS is free again
        if (chunk + NS - 1 < nchunks && !(EXP & 1)) issue_dma((chunk + NS - 1) % NS);
        compute(chunk, b);
    }

    // ---- output transform A^T M A per lane: tiles 4 lk + e (e = 0..3) of each tile row, output channel li ----
    float* s_red = s_aux + 3 * cap;
    const int co = li;
    const bool co_ok = co < p.cout;
    const float bias = (co_ok && p.bias) ? p.bias[co] : 0.f;
    const int px = x0 + 8 * lk;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int y = y0 + 2 * (wave * R + r);
        f32x4 row0[2], row1[2];          // 8 consecutive pixels of output rows y and y + 1
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float u0[4], u1[4];          // A^T M: rows (m0 + m1 + m2, m1 - m2 - m3), per column
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float m0 = acc[r][c][e], m1 = acc[r][4 + c][e], m2 = acc[r][8 + c][e], m3 = acc[r][12 + c][e];
                u0[c] = m0 + m1 + m2;
                u1[c] = m1 - m2 - m3;
            }
            row0[e >> 1][2 * (e & 1)] = u0[0] + u0[1] + u0[2] + bias;
            row0[e >> 1][2 * (e & 1) + 1] = u0[1] - u0[2] - u0[3] + bias;
            row1[e >> 1][2 * (e & 1)] = u1[0] + u1[1] + u1[2] + bias;
            row1[e >> 1][2 * (e & 1) + 1] = u1[1] - u1[2] - u1[3] + bias;
        }
        if (co_ok) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (px + 4 * half + 3 < p.w) {
                    float* dst = p.out + n * p.out_ns + static_cast<int64_t>(co) * p.out_cs + static_cast<int64_t>(y) * p.out_w + px + 4 * half;
                    if (y < p.h) {
                        *reinterpret_cast<f32x4*>(dst) = row0[half];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s1 += row0[half][e]; s2 += row0[half][e] * row0[half][e]; }
                    }
                    if (y + 1 < p.h) {
                        *reinterpret_cast<f32x4*>(dst + p.out_w) = row1[half];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s1 += row1[half][e]; s2 += row1[half][e] * row1[half][e]; }
                    }
                }
            }
        }
    }
    if (p.out_sums) {
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        if (lk == 0) {
            s_red[(wave * 16 + li) * 2] = s1;
            s_red[(wave * 16 + li) * 2 + 1] = s2;
        }
        __syncthreads();
        if (tid < 32) {
            const int j = tid >> 1, which = tid & 1;
            if (j < p.cout) {
                double t = 0.0;
                for (int wv = 0; wv < 4; ++wv) t += static_cast<double>(s_red[(wv * 16 + j) * 2 + which]);
                atomicAdd(p.out_sums + 2 * j + which, t);
            }
        }
    }
}

inline bool wino_fwd_ok(const ConvParams& p) {
    return p.cout <= 16 && (p.w % 4 == 0) && (p.in_w % 4 == 0) && (p.out_w % 4 == 0) && (p.in_cs % 4 == 0) && (p.in_ns % 4 == 0) &&
           (p.out_cs % 4 == 0) && (p.out_ns % 4 == 0) && (reinterpret_cast<uintptr_t>(p.in) % 16 == 0) &&
           (reinterpret_cast<uintptr_t>(p.out) % 16 == 0) && (reinterpret_cast<uintptr_t>(p.wgt) % 16 == 0) && p.ksplit == 0 &&
           p.cin % 4 == 0 && p.cin >= 8;          // whole K-chunks only (dense layers: cin = 48 + 12 j)
}

template <int R, int KC, int MINW, int NS, int EXP = 0>
inline int launch_wino_fwd(ConvParams p, hipStream_t stream) {
    using G = WinoFwdGeom<R, KC, NS>;
    p.tiles_x = (p.w + G::kTileX - 1) / G::kTileX;
    p.bn_cap = ((p.cin + KC - 1) / KC * KC + 15) / 16 * 16;
    const int tiles_y = (p.h + G::kTileY - 1) / G::kTileY;
    const size_t smem = G::bytes(p.bn_cap);
    static size_t configured_by_device[16] = {};          // the attribute belongs to the (function, device) pair
    int dev = 0;
    (void)hipGetDevice(&dev);
    size_t& configured = configured_by_device[dev & 15];
    if (smem > 48 * 1024 && smem > configured) {
        ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fwd_kernel<R, KC, MINW, NS, EXP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(smem)));
        configured = smem;
    }
    wino_fwd_kernel<R, KC, MINW, NS, EXP><<<dim3(p.tiles_x * tiles_y, 1, p.n), kConvThreads, smem, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}


}  // namespace endo
