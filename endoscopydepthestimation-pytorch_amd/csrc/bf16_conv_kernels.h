// bf16-STORAGE kernel family, first bricks (round 3; BASELINE configs[2] / [4]): convolutions over channels-last bf16
// level buffers on the bf16 matrix cores.
//
// Why another family and another layout.  fp32 matrix instructions run on the vector FMA lanes of this part, so the
// fp32 path is bound by the SUM of its matrix and vector work; v_mfma_f32_16x16x32_bf16 is a separate unit at 16x the rate.  With
// bf16 operands a lane supplies 8 consecutive k per instruction, and for a convolution k runs over INPUT CHANNELS -- which the
// planar (NCHW) buffers of the fp32 family put a whole plane apart.  Here a level buffer is [n][h][w][t] bf16 (NHWC: the t channels
// of a pixel are contiguous): a lane's 8 k are one 16-byte read, a dense layer reads one contiguous run of Cin channels per pixel
// and appends its 12 outputs as 24 contiguous bytes.
//
// Channel blocks.  A buffer is [n][t / blk][h][w][blk]: blk = t is plain NHWC; the level buffers use blk = 32 (64-byte records, one
// K-chunk of one pixel, neighbouring pixels adjacent): with NHWC a K-chunk touches 64 bytes of every 384-byte pixel record, the other
// half of each 128-byte line is fetched again by the next chunk after the L2 has dropped it -- measured 2.1x the algorithmic bytes
// at level 0 (profiles/r03_p_bf16_conv_layout.txt); blocked, a chunk of a tile row is one contiguous run.
//
// MFMA roles (v_mfma_f32_16x16x32_bf16, fp32 accumulation): A[i = cout][k = channel] = weights, B[k = channel][j = pixel] =
// activations, D[i = cout][j = pixel]: a lane ends up with 4 consecutive output channels of ONE pixel -- an 8-byte NHWC store.
// A block owns a 16 x 32 pixel tile (WAVES waves x 16 / WAVES rows); per K-chunk of 32 input channels the haloed tile (18 x 34 pixels x 64 B)
// is staged through registers into LDS with BatchNorm + ReLU applied ONCE per element on the way (the fp32 kernels apply it on
// every fragment read; here each staged element is read for 9 taps) and out-of-image pixels written as the zeros of the padded
// post-activation tensor (reference models.py:22-25 pads relu(bn(x))).  16-byte slots are XOR-swizzled by (x >> 1) & 3 so that the
// fragment reads (one ds_read_b128 per lane: 16 lanes x 64-byte stride) are conflict-free.  A fragment read at LDS row r serves
// the three output rows r, r - 1, r - 2 (ky taps): 36 fragment + 9 weight reads feed 72 MFMAs per wave and chunk.
//
// Per-pixel budget of the widest dense layer (level 0, Cin = 180), in cycles of one SIMD: HBM 106 (360 B at 8 TB/s over 1024 SIMDs),
// matrix 54, vector 39 (unpack / fma / max / pack once per element) -- HBM-bound, ~5x under the fp32 Winograd kernel's 845.
#pragma once

#include "common.h"

// ELEMENT TYPE.  The family is written once for a 16-bit storage type "s16": bf16 (net16.hip; BASELINE configs[2]) or, compiled a second
// time with ENDO16_HALF (net16h.hip; configs[4]'s "fp16 storage / fp32 accumulate"), IEEE half.  What differs: the two conversions, the
// matrix instruction (v_mfma_f32_16x16x32_bf16 / _f16), the stochastic-rounding step, and -- half only -- a power-of-two gradient scale
// (the per-pixel gradients of a mean loss are ~1e-6, below half's smallest normal 6e-5).  Each build lives in its own inline namespace.
#ifdef ENDO16_HALF
#define ENDO16_NS h16
#define S16_MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
#define ENDO16_NS b16
#define S16_MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif

namespace endo {
inline namespace ENDO16_NS {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

constexpr int kBfTileX = 32, kBfTileY = 16;          // output pixels per block
constexpr int kBfKC = 32;                            // input channels per K-chunk (one MFMA k extent)


struct Conv16Params {
    int n, h, w;                     // output grid
    const uint16_t* in;              // bf16 NHWC input level buffer
    int64_t in_ns;                   // elements between samples
    int in_t;                        // channels of the input buffer
    int in_blk;                      // channels per block of the input buffer (0 = in_t: plain channels-last); see the layout note above
    int in_h, in_w;                  // input grid (== h, w unless `ups`)
    int ic0, cin;                    // the layer reads channels [ic0, ic0 + cin)
    const float* bn;                 // [cin][2] = (scale, shift): z = max(x * scale + shift, 0); null = raw input (unless in_sums)
    // BatchNorm of the input from statistics instead of a table (reference models.py:22: nn.BatchNorm2d in front of every dense /
    // transition-down convolution): training != 0: batch statistics from in_sums ([cin][2] fp64 sum, sum^2 over `count` values per
    // channel), running statistics updated (momentum, unbiased variance) and (mean, rstd) saved by the first block; 0: running
    // statistics.  scale = gamma * rstd, shift = beta - mean * scale.
    const double* in_sums;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float* saved;                    // [cin][2] (mean, rstd) or null
    double count;
    float eps, momentum;
    int training;
    int use_stats;                   // 1: the block above applies (bn is ignored)
    // BatchNorm parameter (gamma, beta, running, saved) of the layer's input channel c < rot_n is (c + rot) % rot_n: the up path keeps
    // [skip | transition-up output] in the buffer where the reference concatenates [transition-up output, skip] (models.py:183)
    int rot, rot_n;
    const uint16_t* wgt;             // [chunk][tap][nt][16 cout][32 k] bf16, k >= cin and cout >= `cout` zero
    const float* bias;               // [cout] or null
    uint16_t* out;                   // bf16 NHWC output level buffer
    int64_t out_ns;
    int out_t;
    int out_blk;                     // channels per block of the output buffer (0 = out_t)
    int oc0, cout;
    double* out_sums;                // [cout][2] sum, sum^2 of the STORED (bf16-rounded) values, or null
    int ups;                         // 1: nearest x2 upsampling of the input on the way in (transition up, models.py:73)
    // POOL epilogue (transition down, models.py:64: MaxPool2d(2)): out is the (h / 2) x (w / 2) grid; out_idx: one byte per pooled
    // value, [n][h / 2][w / 2][cout], the position 2 dy + dx of the maximum inside its window (first maximum in row-major order)
    uint8_t* out_idx;
    // ---- backward use (EPI 2 / 3, UNPOOL): the convolution runs over a GRADIENT buffer with the transposed, flipped weights ----
    // EPI_DGRAD_BN (dense layer / transition down, reference models.py:22-25 differentiated): out is the gradient buffer of the layer's
    // INPUT channels, x their forward values (same geometry and channel positions as out); da = [x * scale + shift > 0] * dz;
    // out += scale * da (read-modify-write); out_sums[ci] += (sum da, sum da * x).  scale / shift of output channel ci come from
    // x_saved / gamma / beta at parameter index (ci + rot) % rot_n (as in the forward pass, bit for bit).
    const uint16_t* x;
    const float* x_saved;            // [cin of the layer][2] (mean, rstd)
    // a SUB-RANGE of the layer's input channels (the dense block's new maps; its base channels go through bf16_dgrad_block_kernel):
    // output channel co of this launch is input channel co_off + co of the layer (weights row, BatchNorm parameters, out_sums index),
    // its weights start at 48-row group grp0 of wgroups (0 = gridDim.y)
    int co_off, grp0, wgroups;
    // SAMPLE GROUPS (the two frames of a training pair in one launch, each with its own BatchNorm batch statistics -- reference
    // train.py:276-277 calls the network twice): sample n belongs to group n / group_n (0 = one group); in_sums, saved / x_saved and
    // out_sums of group g start gs_in_sums, gs_saved, gs_out_sums elements after group 0's.  Running statistics are updated by the
    // launch's first block with group 0's statistics first, then group 1's, as two consecutive calls would.
    int group_n;
    int64_t gs_in_sums, gs_saved, gs_out_sums;
    // copies of out_sums this many doubles apart (common.h kBnSlots; 0 = one copy): a block adds to the copy its index selects -- the
    // BatchNorm-BACKWARD sums of the data-gradient launches, whose thousands of blocks otherwise queue on the same few cache lines; the
    // finalize kernels add the copies up.  The forward statistics keep one copy (every later launch's prologue reads them).
    int64_t out_sums_slot_stride;
    unsigned sr_salt;                // gradient stores: per-launch salt of the stochastic rounding (pack_s16x2_sr)
    // UNPOOL input (transition down backward): `in` is the pooled-resolution gradient (ups = 1 addressing) and a full-resolution pixel
    // takes a channel's value only where in_idx ([n][h / 2][w / 2][cin] bytes, the forward pass's out_idx) names its position
    const uint8_t* in_idx;
};

constexpr int kEpiFwd = 0, kEpiPool = 1, kEpiDgradBn = 2, kEpiSumPool = 3;

#ifdef ENDO16_HALF
typedef _Float16 s16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float s16_lo(unsigned v) { return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(v & 0xffffu))); }
__device__ __forceinline__ float s16_hi(unsigned v) { return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(v >> 16))); }
// two floats -> two halves (round to nearest even) in one dword, low half first
__device__ __forceinline__ unsigned pack_s16x2(float a, float b) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, f16x2_t));
}
#else
typedef bf16x8_t s16x8_t;
__device__ __forceinline__ float s16_lo(unsigned v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float s16_hi(unsigned v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
// two floats -> two bf16 (round to nearest even) in one dword, low half first
__device__ __forceinline__ unsigned pack_s16x2(float a, float b) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
#endif

// Two floats -> two bf16 with STOCHASTIC rounding (16 pseudo-random bits added below the kept mantissa, then truncation): for the
// gradient buffers.  A gradient value is stored, read back and re-stored after something small has been added to it -- another consumer's
// contribution, or the deferred BatchNorm terms, which are 1e-3..1e-4 of the value they correct.  Round-to-nearest returns the old bf16
// value whenever the addend is below half an ulp: the corrections are lost SYSTEMATICALLY (measured at 2 x 256 x 320, training mode:
// 20-60 % of a channel's P x + Q missing, parameter gradients 7e-2 off in relative L2 although every map agreed to 1e-2,
// tests/diag/bf16_pq_check.py).  Stochastic rounding keeps the expectation: coherent sums over pixels -- every parameter gradient is
// one -- see the exact value plus zero-mean noise.  `key` identifies the element pair (position and a per-launch salt): the same inputs
// give the same outputs.
__device__ __forceinline__ unsigned sr_hash(unsigned key) {
    key *= 0x9E3779B1u; key ^= key >> 15; key *= 0x85EBCA77u; key ^= key >> 13;
    return key;
}
#ifdef ENDO16_HALF
// half: dither by (u - 1/2) ulp of the half grid at the value (2^(e - 10), e >= -14), then round to nearest
__device__ __forceinline__ float s16_dither(float v, unsigned r16) {
    int e = static_cast<int>((__builtin_bit_cast(unsigned, v) >> 23) & 0xffu) - 127;
    e = (e < -14 ? -14 : e) - 10;
    const float ulp = __builtin_bit_cast(float, static_cast<unsigned>(e + 127) << 23);
    return fmaf(static_cast<float>(r16) * (1.0f / 65536.0f) - 0.5f, ulp, v);
}
__device__ __forceinline__ unsigned pack_s16x2_sr(float a, float b, unsigned key) {
    const unsigned r = sr_hash(key);
    return pack_s16x2(s16_dither(a, r & 0xffffu), s16_dither(b, r >> 16));
}
#else
__device__ __forceinline__ unsigned pack_s16x2_sr(float a, float b, unsigned key) {
    const unsigned r = sr_hash(key);
    const unsigned ua = __builtin_bit_cast(unsigned, a) + (r & 0xffffu), ub = __builtin_bit_cast(unsigned, b) + (r >> 16);
    return (ua >> 16) | (ub & 0xffff0000u);
}
#endif

// sum over the 16 lanes of a DPP row (the 16 pixels li of one lane group): four v_add with row_shr 8 / 4 / 2 / 1, valid in lane 15 of the
// row.  (__shfl_xor is a ds_bpermute -- an LDS-pipe instruction; the BatchNorm sums need 8 of them per value and a fused
// data-gradient block 768 per wave: measured as the largest single cost of that kernel.)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    return v;
}

// byte offset of the 16-byte slot (channel part 0..3 of the chunk) of tile pixel (row, x) in the staged tile
template <int COLS>
__device__ __forceinline__ int bf_slot(int row, int x, int part) { return ((row * COLS + x) * 4 + (part ^ ((x >> 1) & 3))) * 16; }

// KS = 3 (pad 1) or 1; NT = 16-cout tiles per block (grid.y covers ceil(cout / (16 NT)) of them); EPI: kEpiFwd store, kEpiPool 2 x 2
// max-pool + codes, kEpiDgradBn ReLU / BatchNorm backward into the gradient buffer, kEpiSumPool 2 x 2 sum added into the coarser
// gradient buffer (transition up backward); UNPOOL = 1: max-unpooling of the input on the way in
// WAVES = waves per block (4 or 8): the 16 tile rows are dealt R = 16 / WAVES to a wave; 8 waves halve the registers a thread needs
// for its share of the staged loads.  WPE = waves per SIMD the register allocation must leave room for.
//
// One tile per block.  A persistent version (a few blocks per CU walking the tiles with one continuous (tile, chunk) pipeline, XCD-banded
// tile order, statistics flushed once per block) was built and measured: no faster at equal occupancy and 60 registers over the
// 128-register budget of two 8-wave blocks per CU (profiles/r03_o_bf16_conv_variants.txt); not kept.
// EXP: diagnostic masks of tools/bf16_conv_variants (1 = no matrix phase, 2 = no activation loads, 4 = no BatchNorm arithmetic and LDS stage writes); 0 in the product
template <int KS, int NT, int EPI = 0, int WAVES = 8, int WPE = 4, int EXP = 0, int UNPOOL = 0, int TY = kBfTileY>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WPE))) bf16_conv_kernel(const Conv16Params p) {
    constexpr int kThreads = 64 * WAVES;
    constexpr int R = TY / WAVES;          // TY = tile rows (16; 8 for the 4-wave data-gradient blocks that share a CU with a weight-gradient block)
    constexpr int kHalo = KS / 2;
    constexpr int kRows = TY + 2 * kHalo, kCols = kBfTileX + 2 * kHalo;
    constexpr int kPix = kRows * kCols;
    constexpr int kUnits = kPix * 4;                                   // 16-byte units of a chunk
    constexpr int kIter = (kUnits + kThreads - 1) / kThreads;
    constexpr int kTaps = KS * KS;
    constexpr int kWUnits = kTaps * NT * 16 * 4;                       // 16-byte units of a chunk's weight slice
    constexpr int kWIter = (kWUnits + kThreads - 1) / kThreads;
    constexpr bool QUADS = EPI == kEpiDgradBn && UNPOOL == 0;          // the input run starts at a multiple of 4 channels, not of 8
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    unsigned char* s_act = smem16;                                     // [kRows][kCols][4 slots][16 B]
    unsigned char* s_w = s_act + kPix * 64;                            // [tap][nt][16 cout][4 slots][16 B]
    float* s_bn = reinterpret_cast<float*>(s_w + kWUnits * 16);        // [max(cin padded to 32, NT * 16)][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int tiles_x = (p.w + kBfTileX - 1) / kBfTileX, tiles_y = (p.h + TY - 1) / TY;
    const int tiles_img = tiles_x * tiles_y;
    const int co_base = blockIdx.y * NT * 16;
    const int nchunks = (p.cin + kBfKC - 1) / kBfKC;
    float* s_red = s_bn + 2 * (nchunks * kBfKC > NT * 16 ? nchunks * kBfKC : NT * 16);          // [WAVES][NT * 16][2]
    const int in_plane = p.in_h * p.in_w;

    const int grp = p.group_n > 0 ? static_cast<int>(blockIdx.z) / p.group_n : 0;
    const int ngrp = p.group_n > 0 ? static_cast<int>(gridDim.z) / p.group_n : 1;
    const bool first_block = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;                       // of the launch
    const bool first_of_group = blockIdx.x == 0 && blockIdx.y == 0 && static_cast<int>(blockIdx.z) == grp * p.group_n;
    const bool has_bn = EPI != kEpiDgradBn && (p.bn != nullptr || p.use_stats != 0);
    if constexpr (EPI == kEpiDgradBn) {
        // (scale, shift) of this block's NT * 16 output channels = input channels of the differentiated layer
        for (int c = tid; c < NT * 16; c += kThreads) {
            const int ci = p.co_off + co_base + c;
            float sc = 0.f, sh = 0.f;
            if (co_base + c < p.cout) {
                const int pc = ci < p.rot_n ? (ci + p.rot < p.rot_n ? ci + p.rot : ci + p.rot - p.rot_n) : ci;
                const float* sv = p.x_saved + grp * p.gs_saved;
                sc = p.gamma[pc] * sv[2 * pc + 1];
                sh = fmaf(-sv[2 * pc], sc, p.beta[pc]);
            }
            s_bn[2 * c] = sc; s_bn[2 * c + 1] = sh;
        }
    } else
    for (int c = tid; c < nchunks * kBfKC; c += kThreads) {
        float sc = 0.f, sh = 0.f;
        if (c < p.cin) {
            const int pc = c < p.rot_n ? (c + p.rot < p.rot_n ? c + p.rot : c + p.rot - p.rot_n) : c;
            if (p.use_stats) {
                auto stats_of = [&](int g, double& mean, double& var) {
                    const double* sums = p.in_sums + g * p.gs_in_sums;
                    mean = sums[2 * c] / p.count;
                    var = sums[2 * c + 1] / p.count - mean * mean;
                    if (var < 0.0) var = 0.0;
                };
                double mean, var;
                if (p.training) {
                    stats_of(grp, mean, var);
                } else {
                    mean = p.running_mean[pc];
                    var = p.running_var[pc];
                }
                const float rstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(p.eps)));
                const float mean_f = static_cast<float>(mean);
                sc = p.gamma[pc] * rstd;
                sh = fmaf(-mean_f, sc, p.beta[pc]);
                if (first_of_group && p.saved) { float* sv = p.saved + grp * p.gs_saved; sv[2 * pc] = mean_f; sv[2 * pc + 1] = rstd; }
                if (first_block && p.training) {
                    float rm = p.running_mean[pc], rv = p.running_var[pc];
                    for (int g = 0; g < ngrp; ++g) {
                        double gm, gv;
                        stats_of(g, gm, gv);
                        const double unbiased = p.count > 1.0 ? gv * p.count / (p.count - 1.0) : gv;
                        rm = (1.0f - p.momentum) * rm + p.momentum * static_cast<float>(gm);
                        rv = (1.0f - p.momentum) * rv + p.momentum * static_cast<float>(unbiased);
                    }
                    p.running_mean[pc] = rm; p.running_var[pc] = rv;
                }
            } else {
                sc = p.bn ? p.bn[2 * pc] : 1.f; sh = p.bn ? p.bn[2 * pc + 1] : 0.f;
            }
        }
        s_bn[2 * c] = sc; s_bn[2 * c + 1] = sh;
    }

    // this thread's units of a chunk: the LDS slot (the same for every tile and chunk, -1 = no unit) and, per tile, the element offset
    // of the unit's pixel record inside its channel block (-1 = outside the image / no unit); the channel part is tid & 3 for all of
    // them (the thread count is a multiple of 4)
    int u_dst[kIter];
    int u_off[kIter];
#pragma unroll
    for (int i = 0; i < kIter; ++i) {
        const int u = tid + i * kThreads;
        u_dst[i] = -1;
        if (u < kUnits) {
            const int px = u >> 2, part = u & 3;
            const int ry = px / kCols, rx = px - ry * kCols;
            u_dst[i] = bf_slot<kCols>(ry, rx, part);
        }
    }
    auto tile_offsets = [&](int y0, int x0, int (&off)[kIter]) {
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            const int px = (tid + i * kThreads) >> 2;
            const int ry = (px * 241) >> 13 /* px / 34 for px < 612 */, rx = px - ry * kCols;
            static_assert(kCols == 34 || kCols == 32, "tile pixel row from the unit index");
            const int ry_ = kCols == 34 ? ry : px >> 5, rx_ = kCols == 34 ? rx : px & 31;
            const int gy = y0 - kHalo + ry_, gx = x0 - kHalo + rx_;
            const int sy = p.ups ? gy >> 1 : gy, sx = p.ups ? gx >> 1 : gx;
            off[i] = (u_dst[i] < 0 || gy < 0 || gy >= p.h || gx < 0 || gx >= p.w) ? -1 : (sy * p.in_w + sx) * p.in_blk;
        }
    };

    u32x4_t raw[kIter];
    u32x2_t craw[UNPOOL ? kIter : 1];
    u32x4_t wraw[kWIter];
    auto issue_loads = [&](int chunk, int n, const int (&off)[kIter]) {
        // the thread's 8 channels of this chunk: block and position inside the block
        const int cpart = chunk * kBfKC + (tid & 3) * 8;
        const int cb = (p.ic0 + cpart) / p.in_blk;
        const uint16_t* base = p.in + n * p.in_ns + static_cast<int64_t>(cb) * in_plane * p.in_blk + (p.ic0 + cpart - cb * p.in_blk);
        if constexpr (QUADS) {
            // ic0 is a multiple of 4 only (a dense layer's 12 maps start at multiples of 12): the unit is read as two 4-channel halves,
            // each inside one channel block
            const int cb1 = (p.ic0 + cpart + 4) / p.in_blk;
            const uint16_t* base1 = p.in + n * p.in_ns + static_cast<int64_t>(cb1) * in_plane * p.in_blk + (p.ic0 + cpart + 4 - cb1 * p.in_blk);
#pragma unroll
            for (int i = 0; i < kIter; ++i) {
                raw[i] = u32x4_t{0u, 0u, 0u, 0u};
                if (off[i] >= 0) {
                    if (cpart < p.cin) { const u32x2_t h0 = *reinterpret_cast<const u32x2_t*>(base + off[i]); raw[i][0] = h0[0]; raw[i][1] = h0[1]; }
                    if (cpart + 4 < p.cin) { const u32x2_t h1 = *reinterpret_cast<const u32x2_t*>(base1 + off[i]); raw[i][2] = h1[0]; raw[i][3] = h1[1]; }
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            raw[i] = u32x4_t{0u, 0u, 0u, 0u};
            if ((EXP & 2) == 0 && off[i] >= 0) {
                // the last unit of the last chunk may straddle cin: it is read as far as the layer goes and the rest stays zero
                const uint16_t* src = base + off[i];
                if constexpr (UNPOOL != 0) {          // cin is a multiple of 8 here; off / in_blk = the pooled pixel
                    craw[i] = u32x2_t{0xffffffffu, 0xffffffffu};
                    if (cpart + 8 <= p.cin)
                        craw[i] = *reinterpret_cast<const u32x2_t*>(p.in_idx + (static_cast<int64_t>(n) * in_plane + off[i] / p.in_blk) * p.cin + cpart);
                }
                if (cpart + 8 <= p.cin) raw[i] = *reinterpret_cast<const u32x4_t*>(src);
                else if (cpart < p.cin) {          // cin is a multiple of 4: the unit holds 4 valid channels
                    const u32x2_t half = *reinterpret_cast<const u32x2_t*>(src);
                    raw[i] = u32x4_t{half[0], half[1], 0u, 0u};
                }
            }
        }
        const uint16_t* wsrc = p.wgt + (static_cast<int64_t>(chunk) * (p.wgroups > 0 ? p.wgroups : gridDim.y) + blockIdx.y + p.grp0) * (kWUnits * 8);
#pragma unroll
        for (int i = 0; i < kWIter; ++i) {
            const int u = tid + i * kThreads;
            wraw[i] = u < kWUnits ? *reinterpret_cast<const u32x4_t*>(wsrc + u * 8) : u32x4_t{0u, 0u, 0u, 0u};
        }
    };
    auto write_stage = [&](int chunk) {
        float sc[8], sh[8];
        if (has_bn) {
            const float* cst = s_bn + 2 * (chunk * kBfKC + (tid & 3) * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4_t q = *reinterpret_cast<const f32x4_t*>(cst + 4 * k);
                sc[2 * k] = q[0]; sh[2 * k] = q[1]; sc[2 * k + 1] = q[2]; sh[2 * k + 1] = q[3];
            }
        }
#pragma unroll
        for (int i = 0; i < kIter; ++i) {
            if (u_dst[i] < 0 || (EXP & 4) != 0) continue;
            u32x4_t v = raw[i];
            if constexpr (UNPOOL != 0) {
                static_assert(UNPOOL == 0 || KS == 1, "max-unpooling is staged for 1 x 1 convolutions");
                const int px = (tid + i * kThreads) >> 2;
                const unsigned pos = (((px >> 5) & 1) << 1) | (px & 1);          // tile origins are even: the pixel's place in its window
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned cw = craw[i][k >> 1] >> (16 * (k & 1));
                    const unsigned keep = (((cw & 0xffu) == pos) ? 0x0000ffffu : 0u) | ((((cw >> 8) & 0xffu) == pos) ? 0xffff0000u : 0u);
                    v[k] &= keep;
                }
            }
            if (has_bn) {
                if (u_off[i] >= 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float z0 = fmaxf(fmaf(s16_lo(v[k]), sc[2 * k], sh[2 * k]), 0.f);
                        const float z1 = fmaxf(fmaf(s16_hi(v[k]), sc[2 * k + 1], sh[2 * k + 1]), 0.f);
                        v[k] = pack_s16x2(z0, z1);
                    }
                } else {
                    v = u32x4_t{0u, 0u, 0u, 0u};          // the zero padding of the post-activation tensor
                }
            }
            *reinterpret_cast<u32x4_t*>(s_act + u_dst[i]) = v;
        }
#pragma unroll
        for (int i = 0; i < kWIter; ++i) {
            const int u = tid + i * kThreads;
            if (u < kWUnits) {
                // global order [tap][nt][cout 16][part 4]; LDS slot swizzled by the cout like the activation tile by x
                const int part = u & 3, co = (u >> 2) & 15, rest = u >> 6;
                *reinterpret_cast<u32x4_t*>(s_w + ((rest * 16 + co) * 4 + (part ^ ((co >> 1) & 3))) * 16) = wraw[i];
            }
        }
    };

    f32x4_t acc[R][2][NT];
    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) { s1[t][i] = 0.f; s2[t][i] = 0.f; }
    constexpr bool POOLED = EPI == kEpiPool || EPI == kEpiSumPool;
    const int64_t out_plane = POOLED ? static_cast<int64_t>(p.h >> 1) * (p.w >> 1) : static_cast<int64_t>(p.h) * p.w;

    __syncthreads();          // s_bn
    const int n = blockIdx.z;
    const int y0 = (blockIdx.x / tiles_x) * TY, x0 = (blockIdx.x % tiles_x) * kBfTileX;
    tile_offsets(y0, x0, u_off);
    issue_loads(0, n, u_off);
    {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[r][hh][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            write_stage(chunk);
            __syncthreads();
            if (chunk + 1 < nchunks) issue_loads(chunk + 1, n, u_off);
            // ---- this wave's R output rows x 32 pixels: LDS row lr (tile row R wave + lr) feeds output rows lr - ky ----
            if constexpr ((EXP & 1) == 0)
#pragma unroll
            for (int lr = 0; lr < R + 2 * kHalo; ++lr) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) {
                        const int lx = 16 * hh + li + kx;
                        const s16x8_t b = *reinterpret_cast<const s16x8_t*>(s_act + bf_slot<kCols>(R * wave + lr, lx, lk));
#pragma unroll
                        for (int ky = 0; ky < KS; ++ky) {
                            const int r = lr - ky;
                            if (r < 0 || r >= R) continue;
#pragma unroll
                            for (int t = 0; t < NT; ++t) {
                                const int wrow = (ky * KS + kx) * NT + t;
                                const s16x8_t a = *reinterpret_cast<const s16x8_t*>(s_w + ((wrow * 16 + li) * 4 + (lk ^ ((li >> 1) & 3))) * 16);
                                acc[r][hh][t] = S16_MFMA(a, b, acc[r][hh][t], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }

        // ---- the tile's outputs: + bias, round to bf16, 8-byte stores (4 consecutive couts of one pixel per lane), statistics of the
        // stored values ----
        uint16_t* out_n = p.out + n * p.out_ns;
        // stochastic-rounding key of a gradient store: position inside the sample + the sample's index INSIDE ITS GROUP (one call with
        // two groups then rounds exactly like two calls)
        const unsigned sr_sample = static_cast<unsigned>(n - grp * (p.group_n > 0 ? p.group_n : 0)) * 0x632BE5ABu;
        auto out_ptr = [&](int64_t pix, int co) {          // 4 consecutive channels never straddle a block (oc0, blk multiples of 4)
            const int ca = p.oc0 + co, cb = ca / p.out_blk;
            return out_n + (cb * out_plane + pix) * p.out_blk + (ca - cb * p.out_blk);
        };
        if constexpr (EPI == kEpiPool) {
            // 2 x 2 max pool of the accumulators: rows 2 rp, 2 rp + 1 of the lane, columns li (even) and li + 1 (the neighbouring lane)
            const int hp = p.h >> 1, wp = p.w >> 1;
#pragma unroll
            for (int rp = 0; rp < R / 2; ++rp)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int yp = (y0 >> 1) + (R / 2) * wave + rp, xp = (x0 + 16 * hh + li) >> 1;
                    const bool pix_ok = yp < hp && xp < wp && (li & 1) == 0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int co = co_base + 16 * t + 4 * lk;
                        float best[4];
                        unsigned code = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float v00 = acc[2 * rp][hh][t][i], v10 = acc[2 * rp + 1][hh][t][i];
                            const float v01 = __shfl_xor(v00, 1, 64), v11 = __shfl_xor(v10, 1, 64);
                            float bst = v00; unsigned c = 0;
                            if (v01 > bst) { bst = v01; c = 1; }
                            if (v10 > bst) { bst = v10; c = 2; }
                            if (v11 > bst) { bst = v11; c = 3; }
                            best[i] = bst; code |= c << (8 * i);
                        }
                        if (co >= p.cout || !pix_ok) continue;
#pragma unroll
                        for (int i = 0; i < 4; ++i) best[i] += p.bias ? p.bias[co + i] : 0.f;
                        const unsigned lo = pack_s16x2(best[0], best[1]), hi = pack_s16x2(best[2], best[3]);
                        const int64_t pix = static_cast<int64_t>(yp) * wp + xp;
                        *reinterpret_cast<u32x2_t*>(out_ptr(pix, co)) = u32x2_t{lo, hi};
                        *reinterpret_cast<unsigned*>(p.out_idx + (static_cast<int64_t>(n) * hp * wp + pix) * p.cout + co) = code;
                        const float q[4] = {s16_lo(lo), s16_hi(lo), s16_lo(hi), s16_hi(hi)};
#pragma unroll
                        for (int i = 0; i < 4; ++i) { s1[t][i] += q[i]; s2[t][i] = fmaf(q[i], q[i], s2[t][i]); }
                    }
                }
        } else if constexpr (EPI == kEpiSumPool) {
            // 2 x 2 sums of the accumulators added into the coarser gradient buffer (nearest x2 upsampling differentiated)
            const int hp = p.h >> 1, wp = p.w >> 1;
#pragma unroll
            for (int rp = 0; rp < R / 2; ++rp)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int yp = (y0 >> 1) + (R / 2) * wave + rp, xp = (x0 + 16 * hh + li) >> 1;
                    const bool pix_ok = yp < hp && xp < wp && (li & 1) == 0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int co = co_base + 16 * t + 4 * lk;
                        float sum[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float v = acc[2 * rp][hh][t][i] + acc[2 * rp + 1][hh][t][i];
                            sum[i] = v + __shfl_xor(v, 1, 64);
                        }
                        if (co >= p.cout || !pix_ok) continue;
#pragma unroll
                        for (int i = 0; i < 4; ++i) s1[t][i] += sum[i];          // out_sums: the pixel sum of what is added (fp32, unrounded)
                        uint16_t* dst = out_ptr(static_cast<int64_t>(yp) * wp + xp, co);
                        const u32x2_t old = *reinterpret_cast<const u32x2_t*>(dst);
                        const unsigned key = (static_cast<unsigned>(dst - out_n) + sr_sample) ^ p.sr_salt;
                        *reinterpret_cast<u32x2_t*>(dst) = u32x2_t{pack_s16x2_sr(s16_lo(old[0]) + sum[0], s16_hi(old[0]) + sum[1], key),
                                                                    pack_s16x2_sr(s16_lo(old[1]) + sum[2], s16_hi(old[1]) + sum[3], key + 2)};
                    }
                }
        } else if constexpr (EPI == kEpiDgradBn) {
            // all of the lane's reads first (forward values and old gradients of its R x 2 x NT quads), then the arithmetic and the
            // stores: with one block of 8 waves per CU the memory-level parallelism has to come from inside the wave
            const uint16_t* x_n = p.x + n * p.out_ns;
            int64_t offs[R][2];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int y = y0 + R * wave + r, x = x0 + 16 * hh + li;
                    offs[r][hh] = (y < p.h && x < p.w) ? static_cast<int64_t>(y) * p.w + x : -1;
                }
            u32x2_t xv[NT][R][2], old[NT][R][2];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co = co_base + 16 * t + 4 * lk;
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        xv[t][r][hh] = u32x2_t{0u, 0u}; old[t][r][hh] = u32x2_t{0u, 0u};
                        if (co < p.cout && offs[r][hh] >= 0) {
                            const int64_t o = out_ptr(offs[r][hh], co) - out_n;
                            xv[t][r][hh] = *reinterpret_cast<const u32x2_t*>(x_n + o);
                            old[t][r][hh] = *reinterpret_cast<const u32x2_t*>(out_n + o);
                        }
                    }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co = co_base + 16 * t + 4 * lk;
                if (co >= p.cout) continue;
                const f32x4_t q0 = *reinterpret_cast<const f32x4_t*>(s_bn + 2 * (16 * t + 4 * lk));
                const f32x4_t q1 = *reinterpret_cast<const f32x4_t*>(s_bn + 2 * (16 * t + 4 * lk) + 4);
                const float sc[4] = {q0[0], q0[2], q1[0], q1[2]}, sh[4] = {q0[1], q0[3], q1[1], q1[3]};
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        if (offs[r][hh] < 0) continue;
                        const u32x2_t xq = xv[t][r][hh], oq = old[t][r][hh];
                        const float xf[4] = {s16_lo(xq[0]), s16_hi(xq[0]), s16_lo(xq[1]), s16_hi(xq[1])};
                        const float of[4] = {s16_lo(oq[0]), s16_hi(oq[0]), s16_lo(oq[1]), s16_hi(oq[1])};
                        float nw[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float da = fmaf(xf[i], sc[i], sh[i]) > 0.f ? acc[r][hh][t][i] : 0.f;
                            s1[t][i] += da; s2[t][i] = fmaf(da, xf[i], s2[t][i]);
                            nw[i] = fmaf(sc[i], da, of[i]);
                        }
                        uint16_t* dst = out_ptr(offs[r][hh], co);
                        const unsigned key = (static_cast<unsigned>(dst - out_n) + sr_sample) ^ p.sr_salt;
                        *reinterpret_cast<u32x2_t*>(dst) = u32x2_t{pack_s16x2_sr(nw[0], nw[1], key), pack_s16x2_sr(nw[2], nw[3], key + 2)};
                    }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int y = y0 + R * wave + r, x = x0 + 16 * hh + li;
                    const bool pix_ok = y < p.h && x < p.w;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int co = co_base + 16 * t + 4 * lk;
                        if (co >= p.cout) continue;          // cout is a multiple of 4
                        float v[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = acc[r][hh][t][i] + (p.bias ? p.bias[co + i] : 0.f);
                        const unsigned lo = pack_s16x2(v[0], v[1]), hi = pack_s16x2(v[2], v[3]);
                        if (pix_ok) {
                            *reinterpret_cast<u32x2_t*>(out_ptr(static_cast<int64_t>(y) * p.w + x, co)) = u32x2_t{lo, hi};
                            const float q[4] = {s16_lo(lo), s16_hi(lo), s16_lo(hi), s16_hi(hi)};
#pragma unroll
                            for (int i = 0; i < 4; ++i) { s1[t][i] += q[i]; s2[t][i] = fmaf(q[i], q[i], s2[t][i]); }
                        }
                    }
                }
        }
    }

    if (p.out_sums) {
        // reduce over the 16 pixels of a lane group (li), then over the waves through LDS, one fp64 atomic per channel, sum and block
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) { s1[t][i] = row16_sum(s1[t][i]); s2[t][i] = row16_sum(s2[t][i]); }
        if (li == 15) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s_red[((wave * NT * 16) + 16 * t + 4 * lk + i) * 2] = s1[t][i];
                    s_red[((wave * NT * 16) + 16 * t + 4 * lk + i) * 2 + 1] = s2[t][i];
                }
        }
        __syncthreads();
        for (int e = tid; e < NT * 16 * 2; e += kThreads) {
            const int ch = e >> 1;
            if (co_base + ch < p.cout) {
                double tsum = 0.0;
                for (int wv = 0; wv < WAVES; ++wv) tsum += static_cast<double>(s_red[(wv * NT * 16 + ch) * 2 + (e & 1)]);
                atomicAdd(p.out_sums + bn_slot_offset(p.out_sums_slot_stride) + grp * p.gs_out_sums + 2 * (p.co_off + co_base + ch) + (e & 1), tsum);
            }
        }
    }
}

template <int KS, int NT, int WAVES, int TY = kBfTileY>
inline size_t bf16_conv_smem(int cin) {
    constexpr int kHalo = KS / 2;
    constexpr int kPix = (TY + 2 * kHalo) * (kBfTileX + 2 * kHalo);
    const int cpad = (cin + kBfKC - 1) / kBfKC * kBfKC;
    const size_t bn = sizeof(float) * 2 * (cpad > NT * 16 ? cpad : NT * 16), red = sizeof(float) * WAVES * NT * 16 * 2;
    return static_cast<size_t>(kPix) * 64 + static_cast<size_t>(KS * KS * NT) * 16 * 64 + bn + red;
}

template <int KS, int NT, int EPI = 0, int WAVES = 8, int WPE = 4, int EXP = 0, int UNPOOL = 0, int TY = kBfTileY>
inline int launch_bf16_conv(const Conv16Params& p_, hipStream_t stream) {
    Conv16Params p = p_;
    if (p.in_blk <= 0) p.in_blk = p.in_t;
    if (p.out_blk <= 0) p.out_blk = p.out_t;
    constexpr bool QUADS = EPI == kEpiDgradBn && UNPOOL == 0;
    if ((p.in_blk & 7) || (p.out_blk & 3) || (p.ic0 & (QUADS ? 3 : 7)) || (p.oc0 & 3) || p.in_t % p.in_blk || p.out_t % p.out_blk) return ENDO_E_BADARG;
    const int tiles = ((p.w + kBfTileX - 1) / kBfTileX) * ((p.h + TY - 1) / TY);
    const int ngroups = (p.cout + NT * 16 - 1) / (NT * 16);
    const size_t smem = bf16_conv_smem<KS, NT, WAVES, TY>(p.cin);
    ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bf16_conv_kernel<KS, NT, EPI, WAVES, WPE, EXP, UNPOOL, TY>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(smem)));
    bf16_conv_kernel<KS, NT, EPI, WAVES, WPE, EXP, UNPOOL, TY><<<dim3(tiles, ngroups, p.n), 64 * WAVES, smem, stream>>>(p);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// ---- weights: W[cout][cin][KS][KS] fp32 -> [chunk][group][tap][nt][16 cout][32 k] bf16 (zero padded) --------------------------------
// (the kernel indexes (chunk * gridDim.y + blockIdx.y) * (taps * NT * 16 * 32) + ((tap * NT + nt) * 16 + cout) * 32 + k)
__global__ void __launch_bounds__(256) bf16_conv_weights_kernel(const float* __restrict__ w, int cout, int cin, int ks, int nt,
                                                                uint16_t* __restrict__ out, int rot, int rot_n) {
    const int taps = ks * ks;
    const int nchunks = (cin + kBfKC - 1) / kBfKC;
    const int ngroups = (cout + nt * 16 - 1) / (nt * 16);
    const int64_t total = static_cast<int64_t>(nchunks) * ngroups * taps * nt * 16 * 32;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int k = e & 31, co16 = (e >> 5) & 15;
        int64_t rest = e >> 9;
        const int t = rest % nt; rest /= nt;
        const int tap = rest % taps; rest /= taps;
        const int grp = rest % ngroups;
        const int chunk = rest / ngroups;
        const int co = (grp * nt + t) * 16 + co16, ci = chunk * kBfKC + k;
        float v = 0.f;
        const int pci = ci < rot_n ? (ci + rot < rot_n ? ci + rot : ci + rot - rot_n) : ci;
        if (co < cout && ci < cin) v = w[(static_cast<int64_t>(co) * cin + pci) * taps + tap];
        out[e] = static_cast<uint16_t>(pack_s16x2(v, 0.f) & 0xffffu);
    }
}

}  // inline namespace
}  // namespace endo
