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
