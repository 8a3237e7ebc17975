// clip_grad_norm_(10.0) + SGD(momentum 0.9) on the flat parameter / gradient / momentum buffers
// (reference train.py:327-328, 202).  HBM-bound: 2 passes over 1.37 M floats.
#include "common.h"

namespace endo {

__global__ void __launch_bounds__(256) sq_norm_kernel(const float* __restrict__ g, double* out, int64_t count, float scale) {
    __shared__ double scratch[4];
    float part[1] = {0.f};
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const float v = g[i] * scale;
        part[0] += v * v;
    }
    block_sum_atomic<1>(part, out, scratch);
}

__global__ void __launch_bounds__(256) sgd_clip_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf,
                                                       double* norm, int64_t count, float lr, float mu, float max_norm,
                                                       float scale, int first, const float* __restrict__ skip_flag) {
    const float total = static_cast<float>(sqrt(norm[0]));
    // the non-finite-loss guard (train.py:317-322), taken on the device: a non-zero flag (any rank's, after the bucket's all-reduce)
    // leaves parameters and momentum untouched -- what the reference's zero_grad() + step() does with torch >= 2.0
    if (skip_flag && skip_flag[0] != 0.f) {
        if (blockIdx.x == 0 && threadIdx.x == 0) norm[1] = static_cast<double>(total);
        return;
    }
    float coef = max_norm / (total + 1.0e-6f);
    coef = coef > 1.0f ? 1.0f : coef;
    if (blockIdx.x == 0 && threadIdx.x == 0) norm[1] = static_cast<double>(total);
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const float gv = g[i] * scale * coef;
        g[i] = gv;
        const float b = first ? gv : mu * buf[i] + gv;
        buf[i] = b;
        p[i] -= lr * b;
    }
}

}  // namespace endo

using namespace endo;

extern "C" int endo_sgd_clip_step(float* params, float* grads, float* momentum, double* norm_out, int64_t count, float lr, float mu,
                                  float max_norm, float grad_scale, int first_step, const float* skip_flag, void* stream_) {
    if (!params || !grads || !momentum || !norm_out || count <= 0) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(kProfOptimizer, stream, 0.0, 4.0 * 6.0 * static_cast<double>(count));
    ENDO_CHECK(hipMemsetAsync(norm_out, 0, 2 * sizeof(double), stream));
    int blocks = static_cast<int>((count + 256 * 8 - 1) / (256 * 8));
    blocks = blocks > 1024 ? 1024 : blocks;
    // (one atomic per block on ONE address, served one after the other at ~8 ns: 128 blocks of 40-odd elements per thread instead of 671)
    sq_norm_kernel<<<blocks > 128 ? 128 : blocks, 256, 0, stream>>>(grads, norm_out, count, grad_scale);
    sgd_clip_kernel<<<blocks, 256, 0, stream>>>(params, grads, momentum, norm_out, count, lr, mu, max_norm, grad_scale, first_step, skip_flag);
    ENDO_LAUNCH_CHECK();
    return 0;
}
