// Shared device helpers for libendo_hip.so (gfx950 only; wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/endo_hip.h"

#define ENDO_CHECK(expr)                                 \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return static_cast<int>(_e); \
    } while (0)

#define ENDO_LAUNCH_CHECK() ENDO_CHECK(hipGetLastError())

namespace endo {

constexpr int kWave = 64;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

// BN-backward sums (sum dz, sum dz * xhat per channel) are added up with fp64 atomics, one per value, block and step of a
// data-gradient kernel.  Atomics on ONE address are served one after the other (~8 ns each on this part: thousands of blocks of
// a full-resolution launch queue up behind each other, and a block's next s_waitcnt vmcnt(0) waits for its own to be
// acknowledged -- 40 % of the new-channel passes' time, tools/nl_bench).  So the sums live in kBnSlots copies, `slot_stride`
// doubles apart (0 = a single copy): a block adds to the copy its index selects, the readers add the copies up.  Eight copies
// bring a full-resolution launch down to a few hundred atomics per address (a few us) and keep the readers' prologue short.
constexpr int kBnSlots = 8;

__device__ __forceinline__ int64_t bn_slot_offset(int64_t slot_stride) {
    return static_cast<int64_t>((blockIdx.x + 11 * blockIdx.z) & (kBnSlots - 1)) * slot_stride;
}

__device__ __forceinline__ double bn_slot_sum(const double* __restrict__ s, int64_t slot_stride) {
    if (slot_stride == 0) return s[0];
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kBnSlots; ++k) t += s[k * slot_stride];
    return t;
}

// bf16 operands for v_mfma_f32_16x16x16_bf16 (the "bf16 operands" mode, ENDO_OPT_MFMA_BF16): four consecutive k values of a lane,
// rounded to nearest even by v_cvt_pk_bf16_f32 (gfx950), two per dword.  One such MFMA replaces four v_mfma_f32_16x16x4_f32 whose
// k-steps hold the same 16 k values: lane group lk carries k = 4 lk + i in both forms.
typedef short bf16x4_bits __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x4_bits pack_bf16x4(float a, float b, float c, float d) {
    // through the compiler's own conversion (one v_cvt_pk_bf16_f32 per pair), NOT inline assembly: the hazard recogniser must see
    // the VALU write to insert the wait states an MFMA needs before it reads those registers
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t lo = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t), hi = __builtin_convertvector(f32x2_t{c, d}, bf16x2_t);
    return __builtin_bit_cast(bf16x4_bits, u32x2_t{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)});
}

// eight consecutive k values of a lane for v_mfma_f32_16x16x32_bf16 (gfx950: twice the k of the x16 form in the same 16-18 cycles,
// tools/mfma_rate_probe)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_acc __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8_t pack_bf16x8(float a, float b, float c, float d, float e, float f, float g, float h) {
    typedef float f32x8_t __attribute__((ext_vector_type(8)));
    return __builtin_convertvector(f32x8_t{a, b, c, d, e, f, g, h}, bf16x8_t);
}

// ---- fp32 products on the bf16 matrix cores: the three-term split ("bf16x3") ---------------------------------------------------------
// fp32 matrix instructions run on the vector FMA lanes of this part (157 TFLOP/s, shared with every VALU instruction of the kernel:
// DESIGN.md 4.12); v_mfma_f32_16x16x32_bf16 is a separate unit with 16x the rate.  An fp32 value is EXACTLY the sum of three bf16 terms
//     v = hi + mid + lo,   hi = bf16(v),  mid = bf16(v - hi),  lo = bf16(v - hi - mid)
// (round to nearest even: each residual is at most half an ulp of the term before, so the three terms carry 8 + 9 + 9 >= 24 significant
// bits; the subtractions are exact in fp32), and a product of two bf16 values is exact in fp32.  So
//     a * b = ah bh + (ah bm + am bh) + (am bm + ah bl + al bh) + [am bl + al bm + al bl],
// where the bracket is at most 2^-23 |a b| (|am| <= 2^-8 |a|, |bl| <= 2^-16 |b|), typically 2^-26 |a b| -- relative to ONE product,
// while the rounding fp32 itself commits when it adds a product to its accumulator is 2^-24 of the (much larger) running sum.
// The kernels issue the six leading products as six bf16 MFMAs into the same fp32 accumulator, smallest first: fp32
// operands, exact products, fp32 accumulation -- at 6 / 16 of the fp32 matrix instructions' issue time and off the vector lanes.
// Splitting costs ~5.5 VALU instructions per element and is done once per fragment that then feeds many MFMAs.
typedef unsigned u32x4_bits __attribute__((ext_vector_type(4)));

struct Bf16x8Split {
    bf16x8_t hi, mid, lo;
};

// two values -> one dword of each term; ra / rb are scratch
__device__ __forceinline__ void split_bf16x3_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
    const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{ra, rb}, bf16x2_t));
    const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xffff0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{sa, sb}, bf16x2_t));
}

__device__ __forceinline__ Bf16x8Split split_bf16x8(float a, float b, float c, float d, float e, float f, float g, float h) {
    unsigned hi[4], mid[4], lo[4];
    split_bf16x3_pair(a, b, hi[0], mid[0], lo[0]);
    split_bf16x3_pair(c, d, hi[1], mid[1], lo[1]);
    split_bf16x3_pair(e, f, hi[2], mid[2], lo[2]);
    split_bf16x3_pair(g, h, hi[3], mid[3], lo[3]);
    return Bf16x8Split{__builtin_bit_cast(bf16x8_t, u32x4_bits{hi[0], hi[1], hi[2], hi[3]}), __builtin_bit_cast(bf16x8_t, u32x4_bits{mid[0], mid[1], mid[2], mid[3]}),
                       __builtin_bit_cast(bf16x8_t, u32x4_bits{lo[0], lo[1], lo[2], lo[3]})};
}

// acc += a * b over the 32 k of the fragments, six products, smallest first
__device__ __forceinline__ f32x4_acc mfma_bf16x3(const Bf16x8Split& a, const Bf16x8Split& b, f32x4_acc acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, acc, 0, 0, 0);
    return acc;
}

// Block-wide sum of K per-thread partials, one fp64 atomic per value per block.
// scratch: K * (blockDim/64) doubles of LDS.  All threads must call.
template <int K>
__device__ __forceinline__ void block_sum_atomic(const float (&part)[K], double* dst, double* scratch) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwave = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double v = wave_sum(static_cast<double>(part[k]));
        if (lane == 0) scratch[k * nwave + wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        double v = 0.0;
        for (int i = 0; i < nwave; ++i) v += scratch[threadIdx.x * nwave + i];
        atomicAdd(dst + threadIdx.x, v);
    }
    __syncthreads();
}

// XCD-aware block -> work-item remap.  Workgroup b is observed to run on XCD b % 8 (8 XCDs, each with a
// private 4 MiB L2; MI355X_MICROARCH.md).  Giving XCD k the contiguous range [k*T/8, (k+1)*T/8) of a
// row-major tile grid puts horizontally / vertically adjacent tiles -- which share halo cache lines --
// behind the same L2.  Purely a locality hint: the map is a bijection on [0, T) for any T.
__device__ __forceinline__ int xcd_remap(int b, int total) {
    const int xcd = b & 7, idx = b >> 3;
    const int q = total >> 3, r = total & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace endo

// geometry.hip / losses.hip, used by head.hip: the public entry points with their reduction tables' memset optional (zero = 0: the caller has
// zeroed them -- the loss head does so for all of its tables with one memset).  Not part of the C ABI.
int endo_depth_scale_fwd_impl(const float* pred, const float* sparse_depth, const float* sparse_mask, float* scaled, float* ratio, double* stats,
                              int n, int hw, float eps, int zero, hipStream_t stream);
int endo_depth_scale_bwd_impl(const float* grad_scaled, const float* grad_ratio, const float* pred, const float* sparse_depth, const double* stats,
                              float* grad_pred, double* work, int n, int hw, float eps, int zero, hipStream_t stream);
int endo_sparse_l1_fwd_impl(const float* flows, const float* flows_hat, const float* mask, float* loss, double* stats, int n, int c, int hw,
                            float eps, int zero, hipStream_t stream);

// geometry.hip, used by head.hip: the fused depth-warp + consistency-loss kernels (endo_warp_consistency), forward (phase 1: memset,
// forward kernel, and -- unless the caller asks for the loss only at the end -- the one-wave finalize) and backward (phase 2).  Not part
// of the C ABI.  zero_grads: 1 = the forward kernel zeroes grad_depth_* and the backward kernel also writes the loss (the stand-alone
// call); 0 = the caller has initialised grad_depth_* (the loss head: its flow terms) and reads the loss between the phases.
int endo_consistency_phase(int phase, const float* depth_1, const float* depth_2, const float* boundaries, const float* t_1_wrt_2,
                           const float* r_1_wrt_2, const float* t_2_wrt_1, const float* r_2_wrt_1, const float* intrinsics, float dcl_weight,
                           float eps, float* loss, float* grad_depth_1, float* grad_depth_2, float* workspace, int n, int h, int w,
                           int zero_grads, hipStream_t stream);

namespace endo {

// live profiling hooks (prof.hip)
struct ProfScope {
    int family;
    hipStream_t stream;
    void* slot;
    ProfScope(int family, hipStream_t stream, double flops, double bytes);
    ~ProfScope();
};

enum ProfFamily {
    kProfConv3x3Dense = 0,   // dense-layer conv3x3 Cin->12 forward (BN+ReLU fused on load)
    kProfConv3x3Up = 1,      // transition-up conv3x3 48->48 forward (nearest x2 fused on load)
    kProfConv1x1Pool = 2,    // transition-down conv1x1 + maxpool forward
    kProfConvFirst = 3,      // first conv 3->48
    kProfConvFinal = 4,      // final conv 192->1 + abs (fwd and bwd)
    kProfDgradDense = 5,     // dense-layer dgrad + BN/ReLU backward
    kProfWgradDense = 6,     // dense-layer wgrad
    kProfDgradOther = 7,     // transition dgrads
    kProfWgradOther = 8,     // transition / first conv wgrads
    kProfSmall = 9,          // BN finalize / dY preparation / misc
    kProfGeometry = 10,      // depth scaling, flow, warp
    kProfLoss = 11,          // loss reductions
    kProfOptimizer = 12,     // clip + SGD
};

}  // namespace endo
