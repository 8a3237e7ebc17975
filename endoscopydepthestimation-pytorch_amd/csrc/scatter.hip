// Sparse SfM scatter on the device (reference utils.py:460-612 get_torch_training_data, restated in
// oracle/scatter.py): project the point cloud into both frames of each pair of a batch, keep the points that are
// visible, clean, inside the image, in front of the camera and on the endoscope mask, and write depth / flow /
// mask planes in the NCHW layout the training step consumes.  HBM-bound, microseconds: the point of the kernel is
// that the point cloud, projections and mask stay resident and 14 of the 16 per-step host->device copies go away.
//
// Collisions: numpy fancy-index assignment lets the HIGHEST point index win; here an atomicMax on a per-pixel
// winner plane decides, then only the winner writes.  All projection arithmetic is fp64 with numpy's evaluation
// order (row-vector times matrix-transpose: a 4-term dot product, left to right), np.round = rint (half to even).
#include "common.h"

namespace endo {

struct ScatterParams {
    const double* points;        // [P][4]
    const double* projections;   // [B][2][3][4]
    const double* extrinsics;    // [B][2][4][4]
    const float* visibility;     // [B][P][2]
    const float* clean;          // [P] or null
    const uint8_t* mask;         // [H][W], 255 = inside
    int n_points, batch, height, width;
    float depth_multiplier;
    int* winner;                 // [2][B][H*W]
    float* depth_masks;          // [2][B][1][H][W]
    float* depths;               // [2][B][1][H][W]
    float* flow_masks;           // [2][B][1][H][W]
    float* flows;                // [2][B][2][H][W]
};

__device__ __forceinline__ double dot4(const double* m, const double* p) {
    // numpy's matmul of a (P,4) by a (4,k) matrix accumulates the four products in index order
    double s = p[0] * m[0];
    s += p[1] * m[1];
    s += p[2] * m[2];
    s += p[3] * m[3];
    return s;
}

// pixel (rounded, as double) of point p in frame `frame` of pair b, and its camera-space depth
__device__ __forceinline__ void project(const ScatterParams& q, int b, int frame, const double* pt, double& u, double& v, double& z) {
    const double* pr = q.projections + (static_cast<int64_t>(b) * 2 + frame) * 12;
    const double w = dot4(pr + 8, pt);
    u = rint(dot4(pr, pt) / w);
    v = rint(dot4(pr + 4, pt) / w);
    const double* ex = q.extrinsics + (static_cast<int64_t>(b) * 2 + frame) * 16;
    z = dot4(ex + 8, pt) / dot4(ex + 12, pt);
}

// -1 when the point is rejected, else the flat pixel index
__device__ __forceinline__ int target_pixel(const ScatterParams& q, int b, int frame, int p, double u, double v, double z) {
    bool keep = q.visibility[(static_cast<int64_t>(b) * q.n_points + p) * 2 + frame] > 0.5f;
    if (q.clean) keep = keep && q.clean[p] > 0.5f;
    keep = keep && u <= q.width - 1 && u >= 0.0 && v <= q.height - 1 && v >= 0.0 && z > 0.0;
    if (!keep) return -1;
    const int loc = static_cast<int>(u) + static_cast<int>(v) * q.width;
    return q.mask[loc] == 255 ? loc : -1;
}

__global__ void scatter_vote_kernel(const ScatterParams q) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y, frame = blockIdx.z;
    if (p >= q.n_points) return;
    const double* pt = q.points + static_cast<int64_t>(p) * 4;
    double u, v, z;
    project(q, b, frame, pt, u, v, z);
    const int loc = target_pixel(q, b, frame, p, u, v, z);
    if (loc >= 0) atomicMax(q.winner + (static_cast<int64_t>(frame) * q.batch + b) * q.height * q.width + loc, p);
}

__global__ void scatter_write_kernel(const ScatterParams q) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y, frame = blockIdx.z;
    if (p >= q.n_points) return;
    const double* pt = q.points + static_cast<int64_t>(p) * 4;
    double u, v, z, uo, vo, zo;
    project(q, b, frame, pt, u, v, z);
    const int loc = target_pixel(q, b, frame, p, u, v, z);
    const int64_t plane = static_cast<int64_t>(q.height) * q.width;
    const int64_t img = static_cast<int64_t>(frame) * q.batch + b;
    if (loc < 0 || q.winner[img * plane + loc] != p) return;
    project(q, b, 1 - frame, pt, uo, vo, zo);
    // float32 arithmetic exactly as the reference: integer differences stored in a float32 plane, then divided
    const float fx = static_cast<float>(uo - u) / static_cast<float>(q.width);
    const float fy = static_cast<float>(vo - v) / static_cast<float>(q.height);
    const bool outlier = fabsf(fx) > 5.0f || fabsf(fy) > 5.0f;
    q.flow_masks[img * plane + loc] = outlier ? 0.f : 1.f;
    q.flows[(img * 2) * plane + loc] = outlier ? 0.f : fx;
    q.flows[(img * 2 + 1) * plane + loc] = outlier ? 0.f : fy;
    q.depths[img * plane + loc] = static_cast<float>(z) * q.depth_multiplier;
    q.depth_masks[img * plane + loc] = 1.f;
}

}  // namespace endo

extern "C" int endo_sparse_scatter(const double* points, int n_points, const double* projections, const double* extrinsics,
                                   const float* visibility, const float* clean, const uint8_t* mask, int batch, int height, int width,
                                   float depth_multiplier, int32_t* winner_scratch, float* depth_masks, float* depths, float* flow_masks,
                                   float* flows, void* stream_) {
    using namespace endo;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_points < 0 || batch <= 0 || height <= 0 || width <= 0) return ENDO_E_BADARG;
    if (!projections || !extrinsics || !mask || !winner_scratch || !depth_masks || !depths || !flow_masks || !flows) return ENDO_E_BADARG;
    if (n_points > 0 && (!points || !visibility)) return ENDO_E_BADARG;      // an empty cloud has no point / visibility storage
    const size_t planes = static_cast<size_t>(2) * batch * height * width;
    ENDO_CHECK(hipMemsetAsync(winner_scratch, 0xFF, planes * sizeof(int32_t), stream));       // -1
    ENDO_CHECK(hipMemsetAsync(depth_masks, 0, planes * sizeof(float), stream));
    ENDO_CHECK(hipMemsetAsync(depths, 0, planes * sizeof(float), stream));
    ENDO_CHECK(hipMemsetAsync(flow_masks, 0, planes * sizeof(float), stream));
    ENDO_CHECK(hipMemsetAsync(flows, 0, 2 * planes * sizeof(float), stream));
    if (n_points == 0) return 0;
    ScatterParams q{points, projections, extrinsics, visibility, clean, mask, n_points, batch, height, width, depth_multiplier,
                    winner_scratch, depth_masks, depths, flow_masks, flows};
    ProfScope prof(kProfGeometry, stream, 0.0, 7.0 * planes * 4.0);
    const dim3 grid((n_points + 127) / 128, batch, 2);
    scatter_vote_kernel<<<grid, 128, 0, stream>>>(q);
    ENDO_LAUNCH_CHECK();
    scatter_write_kernel<<<grid, 128, 0, stream>>>(q);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Coloured point cloud of a depth map (reference utils.py:823-852 point_cloud_from_depth, called per frame by
// evaluate.py:272,340): every `downsampling`-th pixel inside the mask becomes (x, y, z, r, g, b), row-major order.
// Deterministic compaction: one block per image row counts its hits, a one-thread kernel scans the row counts, then one
// block per row writes its points at the row's offset, ordered inside the row by a wave-ballot prefix.  fp32 with the reference's operation order ((w - cx) / fx * z: three
// roundings; the compiler must not contract it into an fma).
// ---------------------------------------------------------------------------------------------
namespace endo {

struct CloudParams {
    const float* depth;        // [H][W]
    const uint8_t* color;      // [H][W][3], B G R as cv2 images are
    const float* mask;         // [H][W]
    const float* k;            // [3][3] intrinsics
    int height, width, downsampling, use_threshold;
    float min_threshold, max_threshold;
    int* row_offsets;          // [H + 1]
    float* points;             // [capacity][6]
    int* count;
};

__device__ __forceinline__ bool cloud_keep(const CloudParams& q, int h, int w) {
    if (h % q.downsampling != 0 || w % q.downsampling != 0) return false;
    const int64_t i = static_cast<int64_t>(h) * q.width + w;
    if (!(q.mask[i] > 0.5f)) return false;
    if (q.use_threshold) {
        const int b = q.color[3 * i], g = q.color[3 * i + 1], r = q.color[3 * i + 2];
        const int mx = max(r, max(g, b)), mn = min(r, min(g, b));
        if (!(static_cast<float>(mx) >= q.max_threshold && static_cast<float>(mn) <= q.min_threshold)) return false;
    }
    return true;
}

__global__ void __launch_bounds__(256) cloud_count_kernel(const CloudParams q) {
    __shared__ int s_cnt[4];
    const int h = blockIdx.x;
    int cnt = 0;
    for (int w = threadIdx.x; w < q.width; w += 256) cnt += cloud_keep(q, h, w) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) q.row_offsets[h + 1] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// in place: row_offsets[h + 1] holds row h's count on entry, the inclusive prefix on exit; row_offsets[0] = 0
__global__ void cloud_scan_kernel(int* row_offsets, int height, int* count) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int acc = 0;
        row_offsets[0] = 0;
        for (int h = 0; h < height; ++h) { acc += row_offsets[h + 1]; row_offsets[h + 1] = acc; }
        *count = acc;
    }
}

__global__ void __launch_bounds__(256) cloud_write_kernel(const CloudParams q) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int h = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = q.row_offsets[h];
    __syncthreads();
    const float fx = q.k[0], cx = q.k[2], fy = q.k[4], cy = q.k[5];
    for (int w0 = 0; w0 < q.width; w0 += 256) {
        const int w = w0 + threadIdx.x;
        const bool keep = w < q.width && cloud_keep(q, h, w);
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int wave_off = 0;
        for (int i = 0; i < wave; ++i) wave_off += s_wave[i];
        const int chunk_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (keep) {
            const int64_t i = static_cast<int64_t>(h) * q.width + w;
            const float z = q.depth[i];
            // separate roundings, as numpy evaluates (w - cx) / fx * z
            const float x = __fmul_rn(__fdiv_rn(__fsub_rn(static_cast<float>(w), cx), fx), z);
            const float y = __fmul_rn(__fdiv_rn(__fsub_rn(static_cast<float>(h), cy), fy), z);
            float* dst = q.points + static_cast<int64_t>(s_base + wave_off + before) * 6;
            dst[0] = x; dst[1] = y; dst[2] = z;
            dst[3] = static_cast<float>(q.color[3 * i + 2]);
            dst[4] = static_cast<float>(q.color[3 * i + 1]);
            dst[5] = static_cast<float>(q.color[3 * i]);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += chunk_total;
        __syncthreads();
    }
}

}  // namespace endo

// ------------------------------------------------------------------------------------------
// relative poses of a batch of frame pairs (reference dataset.py:384-399), one thread per pair:
//   relative = E_1 inv(E_2) in fp64; R_1wrt2 = fp32(relative[:3,:3]); t_1wrt2 = fp32(relative[:3,3] / scale);
//   R_2wrt1 = R_1wrt2^T; t_2wrt1 = -R_1wrt2^T t_1wrt2 formed in fp32 from the fp32 values.
// inv(E_2): Gauss-Jordan with partial pivoting on the 4x4 (the reference calls numpy.linalg.inv = LU with partial pivoting;
// both are accurate to a few ulp of fp64, far inside the fp32 rounding of the outputs).
// ------------------------------------------------------------------------------------------
namespace endo {
__global__ void relative_poses_kernel(const double* __restrict__ ext, int batch, double scale, float* __restrict__ r12, float* __restrict__ t12,
                                      float* __restrict__ r21, float* __restrict__ t21) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double* e1 = ext + static_cast<int64_t>(b) * 32;
    const double* e2 = e1 + 16;
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = e2[i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        double best = fabs(a[col][col]);
        for (int r = col + 1; r < 4; ++r)
            if (fabs(a[r][col]) > best) { best = fabs(a[r][col]); piv = r; }
        if (piv != col)
            for (int j = 0; j < 8; ++j) { const double tmp = a[col][j]; a[col][j] = a[piv][j]; a[piv][j] = tmp; }
        const double inv = 1.0 / a[col][col];
        for (int j = 0; j < 8; ++j) a[col][j] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[col][j];
        }
    }
    float rot[3][3], tr[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 4; ++j) {
            double v = 0.0;
            for (int k = 0; k < 4; ++k) v += e1[i * 4 + k] * a[k][4 + j];
            if (j < 3) rot[i][j] = static_cast<float>(v);
            else tr[i] = static_cast<float>(v / scale);
        }
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { r12[b * 9 + i * 3 + j] = rot[i][j]; r21[b * 9 + i * 3 + j] = rot[j][i]; }
        t12[b * 3 + i] = tr[i];
        // row i of -R^T times t, accumulated left to right in fp32 without contraction (numpy.matmul on float32 operands)
        float acc = __fmul_rn(-rot[0][i], tr[0]);
        acc = __fadd_rn(acc, __fmul_rn(-rot[1][i], tr[1]));
        acc = __fadd_rn(acc, __fmul_rn(-rot[2][i], tr[2]));
        t21[b * 3 + i] = acc;
    }
}
}  // namespace endo

extern "C" int endo_relative_poses(const double* pair_extrinsics, int batch, double scale, float* r_1_wrt_2, float* t_1_wrt_2,
                                   float* r_2_wrt_1, float* t_2_wrt_1, void* stream_) {
    if (!pair_extrinsics || !r_1_wrt_2 || !t_1_wrt_2 || !r_2_wrt_1 || !t_2_wrt_1 || batch <= 0 || !(scale != 0.0)) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    endo::relative_poses_kernel<<<(batch + 63) / 64, 64, 0, stream>>>(pair_extrinsics, batch, scale, r_1_wrt_2, t_1_wrt_2, r_2_wrt_1, t_2_wrt_1);
    ENDO_LAUNCH_CHECK();
    return 0;
}

extern "C" int endo_point_cloud(const float* depth, const uint8_t* color_bgr, const float* mask, const float* intrinsics, int height,
                                int width, int downsampling, int use_threshold, float min_threshold, float max_threshold,
                                int32_t* row_offsets, float* points, int32_t* count_out, void* stream_) {
    using namespace endo;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!depth || !color_bgr || !mask || !intrinsics || !row_offsets || !points || !count_out) return ENDO_E_BADARG;
    if (height <= 0 || width <= 0 || downsampling <= 0) return ENDO_E_BADARG;
    CloudParams q{depth, color_bgr, mask, intrinsics, height, width, downsampling, use_threshold, min_threshold, max_threshold,
                  row_offsets, points, count_out};
    ProfScope prof(kProfGeometry, stream, 0.0, 4.0 * height * width * 3);
    cloud_count_kernel<<<height, 256, 0, stream>>>(q);
    ENDO_LAUNCH_CHECK();
    cloud_scan_kernel<<<1, 64, 0, stream>>>(row_offsets, height, count_out);
    ENDO_LAUNCH_CHECK();
    cloud_write_kernel<<<height, 256, 0, stream>>>(q);
    ENDO_LAUNCH_CHECK();
    return 0;
}
