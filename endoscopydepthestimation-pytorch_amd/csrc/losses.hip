// Self-supervised losses: per-sample masked reductions (wave shuffle -> LDS -> one fp64 atomic per
// block), a one-wave finalize, and elementwise backward kernels.  HBM-bound.
#include "common.h"

namespace endo {

constexpr int kLossThreads = 256;
constexpr int kLossItems = 8;

inline dim3 reduce_grid(int hw, int n) { return dim3((hw + kLossThreads * kLossItems - 1) / (kLossThreads * kLossItems), n); }
inline dim3 apply_grid(int hw, int n) {
    int b = (hw + 255) / 256;
    return dim3(b > 1024 ? 1024 : b, n);
}

// ---- SparseMaskedL1Loss (losses.py:62-66) -------------------------------------------------
__global__ void __launch_bounds__(kLossThreads) sparse_l1_reduce(const float* __restrict__ f, const float* __restrict__ fh,
                                                                 const float* __restrict__ mask, double* stats, int c, int hw) {
    __shared__ double scratch[2 * (kLossThreads / 64)];
    const int n = blockIdx.y;
    const int64_t mbase = static_cast<int64_t>(n) * hw, fbase = mbase * c;
    float part[2] = {0.f, 0.f};
    for (int i = blockIdx.x * kLossThreads * kLossItems + threadIdx.x, k = 0; k < kLossItems && i < hw; ++k, i += kLossThreads) {
        const float m = mask[mbase + i];
        float a = 0.f;
        for (int ch = 0; ch < c; ++ch) {
            const int64_t o = fbase + static_cast<int64_t>(ch) * hw + i;
            a += m * fabsf(f[o] - fh[o]);
        }
        part[0] += a;
        part[1] += m;
    }
    block_sum_atomic<2>(part, stats + 2 * n, scratch);
}

__global__ void sparse_l1_finalize(const double* stats, float* loss, int n, float eps) {
    if (threadIdx.x != 0) return;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += static_cast<float>(stats[2 * i]) / (eps + static_cast<float>(stats[2 * i + 1]));
    *loss = acc / static_cast<float>(n);
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__global__ void __launch_bounds__(256) sparse_l1_bwd_kernel(const float* __restrict__ gloss, const float* __restrict__ f,
                                                            const float* __restrict__ fh, const float* __restrict__ mask,
                                                            const double* __restrict__ stats, float* __restrict__ gf,
                                                            float* __restrict__ gfh, int nsamples, int c, int hw, float eps) {
    const int n = blockIdx.y;
    const float coef = *gloss / static_cast<float>(nsamples) / (eps + static_cast<float>(stats[2 * n + 1]));
    const int64_t mbase = static_cast<int64_t>(n) * hw, fbase = mbase * c;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const float m = mask[mbase + i] * coef;
        for (int ch = 0; ch < c; ++ch) {
            const int64_t o = fbase + static_cast<int64_t>(ch) * hw + i;
            const float s = sgn(f[o] - fh[o]) * m;
            if (gf) gf[o] = s;
            if (gfh) gfh[o] = -s;
        }
    }
}

// ---- NormalizedDistanceLoss (losses.py:122-146) ----------------------------------------------
__global__ void __launch_bounds__(kLossThreads) norm_dist_reduce(const float* __restrict__ d, const float* __restrict__ dw,
                                                                 const float* __restrict__ mask, const float* __restrict__ K,
                                                                 double* stats, int h, int w) {
    __shared__ double scratch[4 * (kLossThreads / 64)];
    const int n = blockIdx.y;
    const int hw = h * w;
    const int64_t base = static_cast<int64_t>(n) * hw;
    const float fx = K[9 * n + 0], fy = K[9 * n + 4], cx = K[9 * n + 2], cy = K[9 * n + 5];
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = blockIdx.x * kLossThreads * kLossItems + threadIdx.x, k = 0; k < kLossItems && i < hw; ++k, i += kLossThreads) {
        const int yy = i / w, xx = i - yy * w;
        const float ax = (static_cast<float>(xx) - cx) / fx;
        const float ay = (static_cast<float>(yy) - cy) / fy;
        const float m = mask[base + i], a = d[base + i], b = dw[base + i];
        part[0] += m * a;
        part[1] += m;
        part[2] += m * fabsf(ax * a - ax * b) + m * fabsf(ay * a - ay * b) + m * fabsf(a - b);
        part[3] += m * (a + fabsf(b));
    }
    block_sum_atomic<4>(part, stats + 4 * n, scratch);
}

__device__ __forceinline__ float norm_dist_den(const double* s, float eps) {
    const float mean_value = static_cast<float>(s[0]) / (eps + static_cast<float>(s[1]));
    return 1.0e-5f * mean_value + static_cast<float>(s[3]);
}

__global__ void norm_dist_finalize(const double* stats, float* loss, int n, float eps) {
    if (threadIdx.x != 0) return;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += 2.0f * static_cast<float>(stats[4 * i + 2]) / norm_dist_den(stats + 4 * i, eps);
    *loss = acc / static_cast<float>(n);
}

__global__ void __launch_bounds__(256) norm_dist_bwd_kernel(const float* __restrict__ gloss, const float* __restrict__ d,
                                                            const float* __restrict__ dw, const float* __restrict__ mask,
                                                            const float* __restrict__ K, const double* __restrict__ stats,
                                                            float* __restrict__ gd, float* __restrict__ gdw, int nsamples,
                                                            int h, int w, float eps) {
    const int n = blockIdx.y;
    const int hw = h * w;
    const int64_t base = static_cast<int64_t>(n) * hw;
    const float fx = K[9 * n + 0], fy = K[9 * n + 4], cx = K[9 * n + 2], cy = K[9 * n + 5];
    const float den = norm_dist_den(stats + 4 * n, eps);
    const float g = *gloss / static_cast<float>(nsamples);
    const float cnum = 2.0f * g / den;                                              // d loss / d num
    const float cden = -2.0f * g * static_cast<float>(stats[4 * n + 2]) / (den * den);   // d loss / d den
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const int yy = i / w, xx = i - yy * w;
        const float ax = (static_cast<float>(xx) - cx) / fx;
        const float ay = (static_cast<float>(yy) - cy) / fy;
        const float m = mask[base + i], a = d[base + i], b = dw[base + i];
        const float t = ax * sgn(ax * a - ax * b) + ay * sgn(ay * a - ay * b) + sgn(a - b);
        if (gd) gd[base + i] = m * (cnum * t + cden);
        if (gdw) gdw[base + i] = m * (-cnum * t + cden * sgn(b));
    }
}

// ---- ScaleInvariantLoss (losses.py:22-32) ----------------------------------------------------
__global__ void __launch_bounds__(kLossThreads) scale_inv_reduce(const float* __restrict__ p, const float* __restrict__ q,
                                                                 const float* __restrict__ b, double* stats, int hw, float eps) {
    __shared__ double scratch[3 * (kLossThreads / 64)];
    const int n = blockIdx.y;
    const int64_t base = static_cast<int64_t>(n) * hw;
    float part[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * kLossThreads * kLossItems + threadIdx.x, k = 0; k < kLossItems && i < hw; ++k, i += kLossThreads) {
        const float m = b[base + i];
        const float r = logf(m * p[base + i] + eps) - logf(m * q[base + i] + eps);
        part[0] += r * r;
        part[1] += r;
        part[2] += m;
    }
    block_sum_atomic<3>(part, stats + 3 * n, scratch);
}

__global__ void scale_inv_finalize(const double* stats, float* loss, int n) {
    if (threadIdx.x != 0) return;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) {
        const float r2 = static_cast<float>(stats[3 * i]), r1 = static_cast<float>(stats[3 * i + 1]), wsum = static_cast<float>(stats[3 * i + 2]);
        acc += r2 / wsum + (r1 * r1) / (wsum * wsum);
    }
    *loss = acc / static_cast<float>(n);
}

__global__ void __launch_bounds__(256) scale_inv_bwd_kernel(const float* __restrict__ gloss, const float* __restrict__ p,
                                                            const float* __restrict__ q, const float* __restrict__ b,
                                                            const double* __restrict__ stats, float* __restrict__ gp,
                                                            float* __restrict__ gq, int nsamples, int hw, float eps) {
    const int n = blockIdx.y;
    const int64_t base = static_cast<int64_t>(n) * hw;
    const float wsum = static_cast<float>(stats[3 * n + 2]);
    const float g = *gloss / static_cast<float>(nsamples);
    const float c2 = 2.0f * g / wsum;                                               // * r
    const float c1 = 2.0f * g * static_cast<float>(stats[3 * n + 1]) / (wsum * wsum);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const float m = b[base + i];
        const float up = m * p[base + i] + eps, uq = m * q[base + i] + eps;
        const float r = logf(up) - logf(uq);
        const float gr = c2 * r + c1;
        if (gp) gp[base + i] = gr * m / up;
        if (gq) gq[base + i] = -gr * m / uq;
    }
}

}  // namespace endo

using namespace endo;

extern "C" int endo_sparse_l1_fwd(const float* flows, const float* flows_hat, const float* mask, float* loss, double* stats, int n,
                                  int c, int hw, float eps, void* stream_) {
    return endo_sparse_l1_fwd_impl(flows, flows_hat, mask, loss, stats, n, c, hw, eps, 1, static_cast<hipStream_t>(stream_));
}

// zero = 0: the caller has zeroed `stats` (the loss head)
int endo_sparse_l1_fwd_impl(const float* flows, const float* flows_hat, const float* mask, float* loss, double* stats, int n, int c, int hw,
                            float eps, int zero, hipStream_t stream) {
    if (!flows || !flows_hat || !mask || !loss || !stats || n <= 0 || c <= 0 || hw <= 0) return ENDO_E_BADARG;
    ProfScope prof(kProfLoss, stream, 0.0, 4.0 * (2.0 * c + 1.0) * n * hw);
    if (zero) ENDO_CHECK(hipMemsetAsync(stats, 0, sizeof(double) * 2 * n, stream));
    sparse_l1_reduce<<<reduce_grid(hw, n), kLossThreads, 0, stream>>>(flows, flows_hat, mask, stats, c, hw);
    sparse_l1_finalize<<<1, 64, 0, stream>>>(stats, loss, n, eps);
    ENDO_LAUNCH_CHECK();
    return 0;
}

extern "C" int endo_sparse_l1_bwd(const float* grad_loss, const float* flows, const float* flows_hat, const float* mask,
                                  const double* stats, float* grad_flows, float* grad_hat, int n, int c, int hw, float eps,
                                  void* stream_) {
    if (!grad_loss || !flows || !flows_hat || !mask || !stats || n <= 0 || c <= 0 || hw <= 0) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(kProfLoss, stream, 0.0, 4.0 * (3.0 * c + 1.0) * n * hw);
    sparse_l1_bwd_kernel<<<apply_grid(hw, n), 256, 0, stream>>>(grad_loss, flows, flows_hat, mask, stats, grad_flows, grad_hat, n,
                                                                  c, hw, eps);
    ENDO_LAUNCH_CHECK();
    return 0;
}

extern "C" int endo_norm_dist_fwd(const float* depth, const float* warped, const float* intersect, const float* K, float* loss,
                                  double* stats, int n, int h, int w, float eps, void* stream_) {
    if (!depth || !warped || !intersect || !K || !loss || !stats || n <= 0 || h <= 0 || w <= 0) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(kProfLoss, stream, 0.0, 4.0 * 3.0 * n * h * w);
    ENDO_CHECK(hipMemsetAsync(stats, 0, sizeof(double) * 4 * n, stream));
    norm_dist_reduce<<<reduce_grid(h * w, n), kLossThreads, 0, stream>>>(depth, warped, intersect, K, stats, h, w);
    norm_dist_finalize<<<1, 64, 0, stream>>>(stats, loss, n, eps);
    ENDO_LAUNCH_CHECK();
    return 0;
}

extern "C" int endo_norm_dist_bwd(const float* grad_loss, const float* depth, const float* warped, const float* intersect,
                                  const float* K, const double* stats, float* grad_depth, float* grad_warped, int n, int h, int w,
                                  float eps, void* stream_) {
    if (!grad_loss || !depth || !warped || !intersect || !K || !stats || n <= 0 || h <= 0 || w <= 0) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(kProfLoss, stream, 0.0, 4.0 * 5.0 * n * h * w);
    norm_dist_bwd_kernel<<<apply_grid(h * w, n), 256, 0, stream>>>(grad_loss, depth, warped, intersect, K, stats, grad_depth,
                                                                    grad_warped, n, h, w, eps);
    ENDO_LAUNCH_CHECK();
    return 0;
}

extern "C" int endo_scale_inv_fwd(const float* pred, const float* goal, const float* boundary, float* loss, double* stats, int n,
                                  int hw, float eps, void* stream_) {
    if (!pred || !goal || !boundary || !loss || !stats || n <= 0 || hw <= 0) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(kProfLoss, stream, 0.0, 4.0 * 3.0 * n * hw);
    ENDO_CHECK(hipMemsetAsync(stats, 0, sizeof(double) * 3 * n, stream));
    scale_inv_reduce<<<reduce_grid(hw, n), kLossThreads, 0, stream>>>(pred, goal, boundary, stats, hw, eps);
    scale_inv_finalize<<<1, 64, 0, stream>>>(stats, loss, n);
    ENDO_LAUNCH_CHECK();
    return 0;
}

extern "C" int endo_scale_inv_bwd(const float* grad_loss, const float* pred, const float* goal, const float* boundary,
                                  const double* stats, float* grad_pred, float* grad_goal, int n, int hw, float eps,
                                  void* stream_) {
    if (!grad_loss || !pred || !goal || !boundary || !stats || n <= 0 || hw <= 0) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(kProfLoss, stream, 0.0, 4.0 * 5.0 * n * hw);
    scale_inv_bwd_kernel<<<apply_grid(hw, n), 256, 0, stream>>>(grad_loss, pred, goal, boundary, stats, grad_pred, grad_goal, n,
                                                                 hw, eps);
    ENDO_LAUNCH_CHECK();
    return 0;
}
