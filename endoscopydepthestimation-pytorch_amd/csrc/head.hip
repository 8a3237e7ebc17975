// The loss head of a training iteration as ONE call: everything between the two network outputs and the gradient that goes back
// into the network (reference train.py:279-315 forward, and its autograd backward) -- depth scaling, flow from depth, boundary
// masking, sparse-flow loss, depth warping both ways, depth-consistency loss, the weighted sum, and the whole backward chain
// down to d loss / d prediction.  It composes the library's own entry points (geometry.hip, losses.hip), so the arithmetic is
// the modules'; what it removes is ~60 autograd nodes and as many host round trips between ~45 small launches: in the traced
// step the GPU sat idle for 0.35 ms there.  The backward half runs unconditionally (it is cheap); the caller still decides on
// the non-finite guard from the loss value before it differentiates the network.
#include "common.h"

namespace endo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// out = a + b + c + d  (c, d may be null)
__global__ void __launch_bounds__(256) head_add_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ c, const float* __restrict__ d, int64_t count) {
    for (int64_t i = (blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x) * 4; i < count; i += static_cast<int64_t>(gridDim.x) * blockDim.x * 4) {
        if (i + 3 < count) {
            f32x4 v = *reinterpret_cast<const f32x4*>(a + i);
            const f32x4 vb = *reinterpret_cast<const f32x4*>(b + i);
            v += vb;
            if (c) v += *reinterpret_cast<const f32x4*>(c + i);
            if (d) v += *reinterpret_cast<const f32x4*>(d + i);
            *reinterpret_cast<f32x4*>(out + i) = v;
        } else {
            for (int64_t k = i; k < count; ++k) out[k] = a[k] + b[k] + (c ? c[k] : 0.f) + (d ? d[k] : 0.f);
        }
    }
}

// losses[0..2] = total, dcl, sfl as train.py:299-315 forms them (fp32), losses[3] = 1 when the total is NaN / Inf (train.py:317) else 0: sfl = w_sfl * 0.5 * (a + b), dcl likewise, total = dcl + sfl;
// up[0] = d total / d (each sparse-flow term), up[1] = d total / d (each consistency term)
// out_j = a_j * mask (per sample, broadcast over a_j's channels) for up to six tensors in ONE launch: blockIdx.z = job.  The arithmetic of
// endo_mask_mul (geometry.hip mask_mul_kernel), which the head used to call once per tensor: six launches of ~5 us each, alone on the chip.
struct MaskMulJobs {
    const float* a[6];
    float* out[6];
    int c[6];
};
__global__ void __launch_bounds__(256) head_mask_mul_kernel(const MaskMulJobs jobs, const float* __restrict__ mask, int hw) {
    const int n = blockIdx.y;
    const float* __restrict__ a = jobs.a[0];
    float* __restrict__ out = jobs.out[0];
    int c = jobs.c[0];
#pragma unroll
    for (int j = 1; j < 6; ++j)
        if (blockIdx.z == j) { a = jobs.a[j]; out = jobs.out[j]; c = jobs.c[j]; }          // (no dynamic index into the kernel argument)
    const int64_t mbase = static_cast<int64_t>(n) * hw;
    const int64_t abase = mbase * c;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const float m = mask[mbase + i];
        for (int k = 0; k < c; ++k) out[abase + static_cast<int64_t>(k) * hw + i] = a[abase + static_cast<int64_t>(k) * hw + i] * m;
    }
}

__global__ void head_combine_kernel(const float* __restrict__ parts, float c_sfl, const float* __restrict__ dcl_weighted, float* __restrict__ losses,
                                    float* __restrict__ up) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float sfl = c_sfl * (parts[0] + parts[1]);
        const float dcl = dcl_weighted[0];          // dcl_weight * 0.5 * (term_1 + term_2), from the fused consistency kernel
        const float total = dcl + sfl;
        losses[0] = total;
        losses[1] = dcl;
        losses[2] = sfl;
        losses[3] = (isnan(total) || isinf(total)) ? 1.f : 0.f;          // the guard of train.py:317, decided on the device
        up[0] = c_sfl;
    }
}

}  // namespace endo

using namespace endo;

// (endo_consistency_phase: geometry.hip, declared in common.h)

// (endo_warp_consistency -- the depth-warp + consistency-loss chain as two fused kernels -- lives in geometry.hip.)

extern "C" int64_t endo_loss_head_workspace_floats(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0) return -1;
    const int64_t p = static_cast<int64_t>(n) * h * w;
    return 34 * p + 128 * n + 256;
}

extern "C" int endo_loss_head(const float* pred_1, const float* pred_2, const float* boundaries, const float* sparse_depths_1,
                              const float* sparse_depths_2, const float* sparse_depth_masks_1, const float* sparse_depth_masks_2,
                              const float* sparse_flows_1, const float* sparse_flows_2, const float* sparse_flow_masks_1,
                              const float* sparse_flow_masks_2, const float* t_1_wrt_2, const float* r_1_wrt_2, const float* t_2_wrt_1,
                              const float* r_2_wrt_1, const float* intrinsics, float sfl_weight, float dcl_weight, float eps,
                              float* losses, float* grad_pred_1, float* grad_pred_2, float* workspace, int n, int h, int w, void* stream_) {
    if (!pred_1 || !pred_2 || !boundaries || !sparse_depths_1 || !sparse_depths_2 || !sparse_depth_masks_1 || !sparse_depth_masks_2 ||
        !sparse_flows_1 || !sparse_flows_2 || !sparse_flow_masks_1 || !sparse_flow_masks_2 || !t_1_wrt_2 || !r_1_wrt_2 || !t_2_wrt_1 ||
        !r_2_wrt_1 || !intrinsics || !losses || !grad_pred_1 || !grad_pred_2 || !workspace || n <= 0 || h <= 0 || w <= 0)
        return ENDO_E_BADARG;
    if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int hw = h * w;
    const int64_t p = static_cast<int64_t>(n) * hw;
    // ---- workspace carving (floats); every plane-sized piece stays 16-byte aligned when p % 4 == 0, head_add_kernel copes otherwise ----
    float* ws = workspace;
    auto take = [&](int64_t count) { float* q = ws; ws += (count + 3) / 4 * 4; return q; };
    float* scaled_1 = take(p);      float* scaled_2 = take(p);
    float* flow_1 = take(2 * p);    float* flow_2 = take(2 * p);          // raw, then masked in place
    float* msf_1 = take(2 * p);     float* msf_2 = take(2 * p);           // sparse flows * boundary
    float* msm_1 = take(p);         float* msm_2 = take(p);               // sparse flow masks * boundary
    float* cons_ws = take(endo_warp_consistency_workspace_floats(n, h, w));          // endo_consistency_phase's own carving (warped, intersect, sums)
    float* g_flow_1 = take(2 * p);  float* g_flow_2 = take(2 * p);        // d / d masked flow, then masked in place = d / d raw flow
    float* g_s1 = take(p);          float* g_s2 = take(p);                // d loss / d scaled depth: the flow terms, then += the consistency terms
    double* dstats = reinterpret_cast<double*>(take(2 * (2 * 8 * n + 2 * n + 2 * 2 * n + 2 * 4 * n)));
    double* ds_stats_1 = dstats;            double* ds_stats_2 = ds_stats_1 + 8 * n;
    double* ds_work_1 = ds_stats_2 + 8 * n; double* ds_work_2 = ds_work_1 + n;
    double* l1_stats_1 = ds_work_2 + n;     double* l1_stats_2 = l1_stats_1 + 2 * n;
    double* nd_stats_1 = l1_stats_2 + 2 * n; double* nd_stats_2 = nd_stats_1 + 4 * n;
    float* parts = take(8);          // sfl_1, sfl_2, dcl_1, dcl_2
    float* up = take(4);             // upstream gradients of the four terms
    float* ratio = take(4);          // depth-scaling's second output (unused by the loss)
    int rc;
#define HEAD(call) do { rc = (call); if (rc) return rc; } while (0)
    // ---- forward (train.py:279-315) ----
    // one memset for the reduction tables of both frames' depth scaling (forward sums, backward work) and sparse-flow losses: 22 n doubles
    ENDO_CHECK(hipMemsetAsync(dstats, 0, sizeof(double) * (2 * 8 * n + 2 * n + 2 * 2 * n), stream));
    HEAD(endo_depth_scale_fwd_impl(pred_1, sparse_depths_1, sparse_depth_masks_1, scaled_1, ratio, ds_stats_1, n, hw, eps, 0, stream));
    HEAD(endo_depth_scale_fwd_impl(pred_2, sparse_depths_2, sparse_depth_masks_2, scaled_2, ratio + 1, ds_stats_2, n, hw, eps, 0, stream));
    HEAD(endo_flow_from_depth_fwd(scaled_1, boundaries, t_1_wrt_2, r_1_wrt_2, intrinsics, flow_1, n, h, w, stream_));
    HEAD(endo_flow_from_depth_fwd(scaled_2, boundaries, t_2_wrt_1, r_2_wrt_1, intrinsics, flow_2, n, h, w, stream_));
    {
        int bx = (hw + 255) / 256;
        bx = bx > 1024 ? 1024 : bx;
        MaskMulJobs jobs{{sparse_flow_masks_1, sparse_flow_masks_2, sparse_flows_1, sparse_flows_2, flow_1, flow_2},
                         {msm_1, msm_2, msf_1, msf_2, flow_1, flow_2}, {1, 1, 2, 2, 2, 2}};
        head_mask_mul_kernel<<<dim3(bx, n, 6), 256, 0, stream>>>(jobs, boundaries, hw);
        ENDO_LAUNCH_CHECK();
    }
    HEAD(endo_sparse_l1_fwd_impl(msf_1, flow_1, msm_1, parts + 0, l1_stats_1, n, 2, hw, 1.0f, 0, stream));
    HEAD(endo_sparse_l1_fwd_impl(msf_2, flow_2, msm_2, parts + 1, l1_stats_2, n, 2, hw, 1.0f, 0, stream));
    // depth warp both ways + depth-consistency loss: the two fused kernels of endo_warp_consistency (geometry.hip); the forward one
    // leaves dcl_weight * 0.5 * (term_1 + term_2) in parts[4] and the backward coefficients in its workspace
    HEAD(endo_consistency_phase(1, scaled_1, scaled_2, boundaries, t_1_wrt_2, r_1_wrt_2, t_2_wrt_1, r_2_wrt_1, intrinsics, dcl_weight, eps,
                                parts + 4, g_s1, g_s2, cons_ws, n, h, w, 0, stream));
    head_combine_kernel<<<1, 64, 0, stream>>>(parts, static_cast<float>(static_cast<double>(sfl_weight) * 0.5), parts + 4, losses, up);
    ENDO_LAUNCH_CHECK();
    // ---- backward ----
    HEAD(endo_sparse_l1_bwd(up + 0, msf_1, flow_1, msm_1, l1_stats_1, nullptr, g_flow_1, n, 2, hw, 1.0f, stream_));
    HEAD(endo_sparse_l1_bwd(up + 0, msf_2, flow_2, msm_2, l1_stats_2, nullptr, g_flow_2, n, 2, hw, 1.0f, stream_));
    {
        int bx = (hw + 255) / 256;
        bx = bx > 1024 ? 1024 : bx;
        MaskMulJobs jobs{{g_flow_1, g_flow_2}, {g_flow_1, g_flow_2}, {2, 2}};
        head_mask_mul_kernel<<<dim3(bx, n, 2), 256, 0, stream>>>(jobs, boundaries, hw);
        ENDO_LAUNCH_CHECK();
    }
    // the flow terms initialise d loss / d scaled depth (every element written), the consistency kernel adds its own
    HEAD(endo_flow_from_depth_bwd(g_flow_1, scaled_1, boundaries, t_1_wrt_2, r_1_wrt_2, intrinsics, g_s1, n, h, w, stream_));
    HEAD(endo_flow_from_depth_bwd(g_flow_2, scaled_2, boundaries, t_2_wrt_1, r_2_wrt_1, intrinsics, g_s2, n, h, w, stream_));
    HEAD(endo_consistency_phase(2, scaled_1, scaled_2, boundaries, t_1_wrt_2, r_1_wrt_2, t_2_wrt_1, r_2_wrt_1, intrinsics, dcl_weight, eps,
                                parts + 4, g_s1, g_s2, cons_ws, n, h, w, 0, stream));
    HEAD(endo_depth_scale_bwd_impl(g_s1, nullptr, pred_1, sparse_depths_1, ds_stats_1, grad_pred_1, ds_work_1, n, hw, eps, 0, stream));
    HEAD(endo_depth_scale_bwd_impl(g_s2, nullptr, pred_2, sparse_depths_2, ds_stats_2, grad_pred_2, ds_work_2, n, hw, eps, 0, stream));
#undef HEAD
    return 0;
}
