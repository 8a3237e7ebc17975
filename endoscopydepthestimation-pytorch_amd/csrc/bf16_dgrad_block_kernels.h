// bf16-STORAGE family: the data gradient of a whole dense block with respect to its BASE channels (the block's input, which all four
// layers read; reference models.py:44-52 differentiated) in ONE pass -- the counterpart of the fp32 family's dgrad_block kernels.
//
// Per layer, the data gradient reads and rewrites every input channel's gradient (2 + 2 bytes) and reads its forward value (2 bytes):
// four passes over the base channels of a block are 24 bytes per channel and pixel against the 24 BYTES PER PIXEL of the layer's own 12
// gradient maps.  Here the gradients of the four layers' outputs (48 consecutive channels of the gradient buffer, all prepared) are
// staged once, the four transposed convolutions run back to back on the matrix cores, each result goes through ITS layer's
// BatchNorm / ReLU backward (the mask and scale differ per layer, the forward value x is the same and is read once), and the sum
// scale_j * da_j is added to the gradient buffer in one read-modify-write: 6 bytes per channel and pixel instead of 24.
//
// K window.  v_mfma_f32_16x16x32_bf16 contracts 32 channels; layer j owns channels 12 j .. 12 j + 11 of the 48.  A lane's B fragment
// is one 16-byte unit (8 channels) of a pixel: the window of layer j is units u0_j .. u0_j + 3 with u0 = {0, 0, 2, 2} (channels 0..31
// / 16..47, which contain 0..11, 12..23 / 24..35, 36..47), and the layer's weights sit at k = 12 j + co - 8 u0_j, zero elsewhere
// (bf16_all_weights_kernel, rows below the block's base count).
//
// A block: 16 x 32 pixels x 48 base channels (grid.y = base channels / 48), 8 waves x 2 rows.  LDS: the staged tile, 18 x 34 pixels x 7
// slots of 16 bytes (6 units + 1 pad: a 112-byte pixel pitch puts the 16 pixels of a fragment read in 16 different bank groups), and
// two weight buffers (layer j + 1 arrives by LDS-DMA while layer j multiplies: the converted weights are stored in the LDS image's own
// order, slot swizzle included).  The BatchNorm sums of a layer leave after its phase (DPP row
// sums over the 16 pixels of a lane group, LDS across the waves, one fp64 atomic per channel and sum).
#pragma once

#include "bf16_conv_kernels.h"

namespace endo {
inline namespace ENDO16_NS {

constexpr int kDbLayers = 4;
constexpr int kDbPitch = 112;                        // bytes per staged pixel
constexpr int kDbWaves = 4;                           // 4-wave blocks over 8-row tiles: one wave per SIMD, so a block fits a CU beside a weight-gradient block
constexpr int kDbTileY = 2 * kDbWaves, kDbThreads = 64 * kDbWaves;
constexpr int kDbRows = kDbTileY + 2, kDbCols = kBfTileX + 2;
constexpr int kDbWBytes = 9 * 3 * 16 * 64;           // one layer's weights of a 48-row group

struct DgradBlock16Params {
    int n, h, w;
    const uint16_t* g;               // gradient buffer: the four layers' prepared output gradients at channels [gc0, gc0 + 48)
    uint16_t* out;                   // the same buffer: base channels [0, c0) are read-modify-written
    const uint16_t* x;               // forward level buffer (same geometry)
    int64_t ns;                      // elements per sample
    int blk;                         // channels per block of the buffers
    int gc0, c0;
    const uint16_t* wgt[kDbLayers];  // data-gradient weights of the layers ([group][tap][nt][16][32] bf16, K window as above)
    const float* saved[kDbLayers];   // (mean, rstd) at parameter index (c + rot) % rot_n
    const float* gamma[kDbLayers];
    const float* beta[kDbLayers];
    double* sums[kDbLayers];         // [cin_j][2] (sum da, sum da * x), accumulated
    int rot, rot_n;
    unsigned sr_salt;                // stochastic rounding of the gradient stores (pack_s16x2_sr)
    int group_n;                     // sample groups (bf16_conv_kernels.h): group g's saved / sums start gs_saved / gs_sums elements later
    int64_t gs_saved, gs_sums;
    int64_t sums_slot_stride;        // copies of the sums this many doubles apart (common.h kBnSlots; 0 = one copy)
};

__global__ void __launch_bounds__(kDbThreads) __attribute__((amdgpu_waves_per_eu(2))) bf16_dgrad_block_kernel(const DgradBlock16Params p) {
    constexpr int R = 2, NT = 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_db[];
    unsigned char* s_g = smem_db;                                             // [10][34][7 slots][16 B]
    unsigned char* s_w = s_g + kDbRows * kDbCols * kDbPitch;                  // [tap][nt][16 rows][4 slots][16 B]: ONE layer (a second buffer would not fit beside the weight gradient's 74 KB)
    float* s_bn = reinterpret_cast<float*>(s_w + kDbWBytes);                  // [4 layers][48][2] (scale, shift)
    float* s_red = s_bn + kDbLayers * 48 * 2;                                 // [2][waves][48][2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int tiles_x = (p.w + kBfTileX - 1) / kBfTileX;
    const int y0 = (blockIdx.x / tiles_x) * kDbTileY, x0 = (blockIdx.x % tiles_x) * kBfTileX;
    const int n = blockIdx.z;
    const int co_base = blockIdx.y * 48;
    const int64_t plane = static_cast<int64_t>(p.h) * p.w;
    const int grp = p.group_n > 0 ? n / p.group_n : 0;

    for (int e = tid; e < kDbLayers * 48; e += kDbThreads) {
        const int j = e / 48, c = e - 48 * j;
        const int ci = co_base + c;
        const int pc = ci < p.rot_n ? (ci + p.rot < p.rot_n ? ci + p.rot : ci + p.rot - p.rot_n) : ci;
        const float* sv = p.saved[j] + grp * p.gs_saved;
        const float sc = p.gamma[j][pc] * sv[2 * pc + 1];
        s_bn[2 * e] = sc; s_bn[2 * e + 1] = fmaf(-sv[2 * pc], sc, p.beta[j][pc]);
    }
    // ---- the 48 gradient channels of the haloed tile: 6 units of 8 channels per pixel ----
    {
        const uint16_t* g_n = p.g + n * p.ns;
        // all of a thread's loads first, then its LDS writes (a load -> store loop would pay the memory latency once per unit)
        constexpr int kGUnits = kDbRows * kDbCols * 6, kGIter = (kGUnits + kDbThreads - 1) / kDbThreads;
        u32x4_t gv[kGIter];
#pragma unroll
        for (int i = 0; i < kGIter; ++i) {
            const int u = tid + i * kDbThreads;
            const int px = u / 6, unit = u - 6 * px;
            const int ry = px / kDbCols, rx = px - ry * kDbCols;
            const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
            gv[i] = u32x4_t{0u, 0u, 0u, 0u};
            if (u < kGUnits && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) {
                const int ca = p.gc0 + 8 * unit, cb = ca / p.blk;
                gv[i] = *reinterpret_cast<const u32x4_t*>(g_n + (cb * plane + static_cast<int64_t>(gy) * p.w + gx) * p.blk + (ca - cb * p.blk));
            }
        }
#pragma unroll
        for (int i = 0; i < kGIter; ++i) {
            const int u = tid + i * kDbThreads;
            const int px = u / 6, unit = u - 6 * px;
            if (u < kGUnits) *reinterpret_cast<u32x4_t*>(s_g + px * kDbPitch + unit * 16) = gv[i];
        }
    }
    // weights of a layer: the group's 27 KB are stored in global memory as the LDS image ([tap][nt][16 rows][4 slots], slot = k / 8
    // XOR (row >> 1) & 3: bf16_all_weights_kernel) and copied by LDS-DMA, 1 KB per wave instruction, 27 of them dealt to the waves
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma_w = [&](int j) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.wgt[j]) + static_cast<int64_t>(blockIdx.y) * kDbWBytes + lane * 16;
        unsigned char* dst = s_w;
#pragma unroll
        for (int k = 0; k < (kDbWBytes / 1024 + kDbWaves - 1) / kDbWaves; ++k) {
            const int chunk = wave + kDbWaves * k;
            if (chunk < kDbWBytes / 1024) __builtin_amdgcn_global_load_lds((gptr_t)(src + chunk * 1024), (lptr_t)(dst + chunk * 1024), 16, 0, 0);
        }
    };
    dma_w(0);

    // ---- the lane's outputs: rows R wave + r, columns 16 hh + li, channels co_base + 16 t + 4 lk .. + 3; forward values read once ----
    const uint16_t* x_n = p.x + n * p.ns;
    uint16_t* out_n = p.out + n * p.ns;
    int pix[R][2];                                   // pixel index in the plane, -1 outside the image
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int y = y0 + R * wave + r, x = x0 + 16 * hh + li;
            pix[r][hh] = (y < p.h && x < p.w) ? y * p.w + x : -1;
        }
    // element offset of the lane's quad of tile t at pixel index px (a sample is far below 2^31 elements)
    auto quad_off = [&](int t, int px) {
        const int ca = co_base + 16 * t + 4 * lk, cb = ca / p.blk;
        return (cb * static_cast<int>(plane) + px) * p.blk + (ca - cb * p.blk);
    };
    u32x2_t xv[NT][R][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                xv[t][r][hh] = pix[r][hh] >= 0 ? *reinterpret_cast<const u32x2_t*>(x_n + quad_off(t, pix[r][hh])) : u32x2_t{0u, 0u};
    float total[NT][R][2][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < 4; ++i) total[t][r][hh][i] = 0.f;

#pragma unroll 1
    for (int j = 0; j < kDbLayers; ++j) {
        __builtin_amdgcn_s_waitcnt(0x0070);          // this wave's DMA of layer j's weights (vmcnt 0) and its LDS writes
        __syncthreads();          // weights of layer j (and, the first time, the tile and the BN table) are in LDS; s_red[(j + 1) & 1] is free
        if (j > 0 && tid < 96) {          // the sums of layer j - 1, written before the barrier
            const float* red = s_red + ((j - 1) & 1) * kDbWaves * 96;
            double tsum = 0.0;
            for (int wv = 0; wv < kDbWaves; ++wv) tsum += static_cast<double>(red[wv * 96 + tid]);
            atomicAdd(p.sums[j - 1] + bn_slot_offset(p.sums_slot_stride) + grp * p.gs_sums + 2 * co_base + tid, tsum);
        }
        const int u0 = j < 2 ? 0 : 2;
        const unsigned char* wj = s_w;
        f32x4_t acc[R][2][NT];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[r][hh][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // tap-outer order: the three weight fragments of a tap are read once and used for the wave's R x 2 pixel groups (with the
        // activation fragment outermost the compiler keeps all 27 weight fragments of the layer in registers -- 108 of them)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                s16x8_t a[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    a[t] = *reinterpret_cast<const s16x8_t*>(wj + ((((ky * 3 + kx) * NT + t) * 16 + li) * 4 + (lk ^ ((li >> 1) & 3))) * 16);
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const s16x8_t b = *reinterpret_cast<const s16x8_t*>(s_g + ((R * wave + r + ky) * kDbCols + 16 * hh + li + kx) * kDbPitch + (u0 + lk) * 16);
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[r][hh][t] = S16_MFMA(a[t], b, acc[r][hh][t], 0, 0, 0);
                    }
            }
        // every wave has read layer j's weights: the next layer's arrive while this one's BatchNorm / ReLU backward runs
        __syncthreads();
        if (j + 1 < kDbLayers) dma_w(j + 1);
        // ---- layer j's BatchNorm / ReLU backward on its share ----
        float s1[NT][4], s2[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>(s_bn + 2 * (48 * j + 16 * t + 4 * lk));
            const f32x4_t q1 = *reinterpret_cast<const f32x4_t*>(s_bn + 2 * (48 * j + 16 * t + 4 * lk) + 4);
            const float sc[4] = {q0[0], q0[2], q1[0], q1[2]}, sh[4] = {q0[1], q0[3], q1[1], q1[3]};
#pragma unroll
            for (int i = 0; i < 4; ++i) { s1[t][i] = 0.f; s2[t][i] = 0.f; }
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    if (pix[r][hh] < 0) continue;
                    const u32x2_t xq = xv[t][r][hh];
                    const float xf[4] = {s16_lo(xq[0]), s16_hi(xq[0]), s16_lo(xq[1]), s16_hi(xq[1])};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float da = fmaf(xf[i], sc[i], sh[i]) > 0.f ? acc[r][hh][t][i] : 0.f;
                        s1[t][i] += da; s2[t][i] = fmaf(da, xf[i], s2[t][i]);
                        total[t][r][hh][i] = fmaf(sc[i], da, total[t][r][hh][i]);
                    }
                }
        }
        float* red = s_red + (j & 1) * kDbWaves * 96;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float r1 = row16_sum(s1[t][i]), r2 = row16_sum(s2[t][i]);
                if (li == 15) {
                    red[wave * 96 + 2 * (16 * t + 4 * lk + i)] = r1;
                    red[wave * 96 + 2 * (16 * t + 4 * lk + i) + 1] = r2;
                }
            }
    }
    __syncthreads();
    if (tid < 96) {
        const float* red = s_red + ((kDbLayers - 1) & 1) * kDbWaves * 96;
        double tsum = 0.0;
        for (int wv = 0; wv < kDbWaves; ++wv) tsum += static_cast<double>(red[wv * 96 + tid]);
        atomicAdd(p.sums[kDbLayers - 1] + bn_slot_offset(p.sums_slot_stride) + grp * p.gs_sums + 2 * co_base + tid, tsum);
    }
    // ---- one read-modify-write of the gradient buffer: all reads, then the sums and the writes ----
    u32x2_t old[NT][R][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                old[t][r][hh] = pix[r][hh] >= 0 ? *reinterpret_cast<const u32x2_t*>(out_n + quad_off(t, pix[r][hh])) : u32x2_t{0u, 0u};
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if (pix[r][hh] < 0) continue;
                const u32x2_t o = old[t][r][hh];
                const float* tt = total[t][r][hh];
                const int qo = quad_off(t, pix[r][hh]);
                const unsigned key = (static_cast<unsigned>(qo) + static_cast<unsigned>(n - grp * (p.group_n > 0 ? p.group_n : 0)) * 0x632BE5ABu) ^ p.sr_salt;
                *reinterpret_cast<u32x2_t*>(out_n + qo) = u32x2_t{pack_s16x2_sr(s16_lo(o[0]) + tt[0], s16_hi(o[0]) + tt[1], key),
                                                                  pack_s16x2_sr(s16_lo(o[1]) + tt[2], s16_hi(o[1]) + tt[3], key + 2)};
            }
}

inline size_t bf16_dgrad_block_smem() {
    return static_cast<size_t>(kDbRows) * kDbCols * kDbPitch + kDbWBytes + sizeof(float) * (kDbLayers * 48 * 2 + 2 * kDbWaves * 96);
}

// c0 a multiple of 48 (every dense block of FC-DenseNet57: 48 k base channels), gc0 a multiple of 8
inline int launch_bf16_dgrad_block(const DgradBlock16Params& p, hipStream_t stream) {
    if (p.c0 <= 0 || (p.c0 % 48) || (p.gc0 & 7) || (p.blk & 7)) return ENDO_E_BADARG;
    const int tiles = ((p.w + kBfTileX - 1) / kBfTileX) * ((p.h + kDbTileY - 1) / kDbTileY);
    const size_t smem = bf16_dgrad_block_smem();
    ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bf16_dgrad_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem)));
    bf16_dgrad_block_kernel<<<dim3(tiles, p.c0 / 48, p.n), kDbThreads, smem, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

}  // inline namespace
}  // namespace endo
