// Per-frame inputs of the contaminated-point filter -- reference utils.py:339-404 (get_clean_point_list), the part that touches
// pixels: for every SfM point and frame, whether the point is visible, projects inside the image and the mask, its camera depth,
// and the brightness there: V of cv2.COLOR_BGR2HSV_FULL (= max(B, G, R)) of cv2.bilateralFilter(img / 255, d, sigmaColor,
// sigmaSpace).  The reference filters every whole frame (2.9 M window evaluations each) to read ~300 pixels of it; here one thread
// evaluates the filter at the one pixel its point needs.  The histogram thresholds that follow are host statistics over a few
// hundred numbers (reader.py), as in the reference.
#include "common.h"

namespace endo {

__device__ __forceinline__ int reflect101(int i, int n) {          // cv2 BORDER_REFLECT_101 (BORDER_DEFAULT)
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

__global__ void __launch_bounds__(128) point_brightness_kernel(const uint8_t* __restrict__ imgs, int frames, int height, int width,
                                                              const double* __restrict__ points, int n_points,
                                                              const double* __restrict__ projections, const double* __restrict__ extrinsics,
                                                              const float* __restrict__ visibility, const uint8_t* __restrict__ mask, int radius,
                                                              double color_coeff, double space_coeff, int32_t* __restrict__ valid,
                                                              double* __restrict__ depth, float* __restrict__ brightness) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (p >= n_points) return;
    const int64_t o = static_cast<int64_t>(f) * n_points + p;
    valid[o] = 0; depth[o] = 0.0; brightness[o] = 0.f;
    if (!(visibility[static_cast<int64_t>(p) * frames + f] > 0.5f)) return;
    const double* X = points + 4 * static_cast<int64_t>(p);
    const double* P = projections + 12 * static_cast<int64_t>(f);
    const double* E = extrinsics + 16 * static_cast<int64_t>(f);
    double c[4], q[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = ((E[4 * i] * X[0] + E[4 * i + 1] * X[1]) + E[4 * i + 2] * X[2]) + E[4 * i + 3] * X[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] = ((P[4 * i] * X[0] + P[4 * i + 1] * X[1]) + P[4 * i + 2] * X[2]) + P[4 * i + 3] * X[3];
    const double z = c[2] / c[3];
    const double u = q[0] / q[2], v = q[1] / q[2];
    if (!(u <= width - 1 && u >= 0.0 && v <= height - 1 && v >= 0.0 && z > 0.0)) return;
    const int x = static_cast<int>(rint(u)), y = static_cast<int>(rint(v));          // np.round: half to even
    if (mask[y * width + x] != 255) return;
    const uint8_t* img = imgs + static_cast<int64_t>(f) * height * width * 3;
    const uint8_t* centre = img + (static_cast<int64_t>(y) * width + x) * 3;
    const float c0 = centre[0] / 255.0f, c1 = centre[1] / 255.0f, c2 = centre[2] / 255.0f;
    double num0 = 0.0, num1 = 0.0, num2 = 0.0, den = 0.0;
    for (int i = -radius; i <= radius; ++i)
        for (int j = -radius; j <= radius; ++j) {
            const double rr = sqrt(static_cast<double>(i * i + j * j));
            if (rr > radius) continue;
            const uint8_t* nb = img + (static_cast<int64_t>(reflect101(y + i, height)) * width + reflect101(x + j, width)) * 3;
            const float n0 = nb[0] / 255.0f, n1 = nb[1] / 255.0f, n2 = nb[2] / 255.0f;
            const double l1 = (fabs(static_cast<double>(n0) - c0) + fabs(static_cast<double>(n1) - c1)) + fabs(static_cast<double>(n2) - c2);
            const double w = exp(rr * rr * space_coeff) * exp(l1 * l1 * color_coeff);
            num0 += w * n0; num1 += w * n1; num2 += w * n2;
            den += w;
        }
    const float b0 = static_cast<float>(num0 / den), b1 = static_cast<float>(num1 / den), b2 = static_cast<float>(num2 / den);
    valid[o] = 1;
    depth[o] = z;
    brightness[o] = fmaxf(b0, fmaxf(b1, b2));
}

// cv2.COLOR_BGR2HSV_FULL / COLOR_RGB2HSV_FULL on 8-bit pixels (reference utils.py:449-450, 80-81; dataset.py:434-442), OpenCV's scalar
// fixed-point path: v = max, s = (diff * sdiv[v] + 2048) >> 12, h = (hterm * hdiv[diff] + 2048) >> 12 (+ 256 when negative) with
// sdiv[i] = round((255 << 12) / i), hdiv[i] = round((256 << 12) / (6 i)) (round half to even).  One thread per pixel; writes the uint8
// HWC image and / or the Normalize(0.5, 0.5) fp32 CHW tensor of dataset.py:446-451.
__global__ void __launch_bounds__(256) hsv_full_kernel(const uint8_t* __restrict__ src, int64_t pixels, int blue_index, uint8_t* __restrict__ out_u8,
                                                       float* __restrict__ out_f32) {
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < pixels; i += 256ll * gridDim.x) {
        const int c0 = src[3 * i], c1 = src[3 * i + 1], c2 = src[3 * i + 2];
        const int b = blue_index == 0 ? c0 : c2, g = c1, r = blue_index == 0 ? c2 : c0;
        const int v = max(max(b, g), r), vmin = min(min(b, g), r), diff = v - vmin;
        const int sdiv = v ? static_cast<int>(rint(static_cast<double>(255 << 12) / v)) : 0;
        const int hdiv = diff ? static_cast<int>(rint(static_cast<double>(256 << 12) / (6.0 * diff))) : 0;
        const int s = (diff * sdiv + (1 << 11)) >> 12;
        const int hterm = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
        int h = (hterm * hdiv + (1 << 11)) >> 12;
        h += h < 0 ? 256 : 0;
        h = h > 255 ? 255 : h;
        if (out_u8) { out_u8[3 * i] = static_cast<uint8_t>(h); out_u8[3 * i + 1] = static_cast<uint8_t>(s); out_u8[3 * i + 2] = static_cast<uint8_t>(v); }
        if (out_f32) {          // (x / 255 - 0.5) / 0.5 as albumentations.Normalize(mean 0.5, std 0.5, max_pixel_value 255) evaluates it in fp32
            out_f32[i] = (static_cast<float>(h) - 127.5f) * (1.0f / 127.5f);
            out_f32[pixels + i] = (static_cast<float>(s) - 127.5f) * (1.0f / 127.5f);
            out_f32[2 * pixels + i] = (static_cast<float>(v) - 127.5f) * (1.0f / 127.5f);
        }
    }
}

}  // namespace endo

using namespace endo;

extern "C" int endo_hsv_full(const uint8_t* src, int64_t pixels, int blue_index, uint8_t* out_u8, float* out_f32, void* stream_) {
    if (!src || pixels <= 0 || (blue_index != 0 && blue_index != 2) || (!out_u8 && !out_f32)) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int blocks = static_cast<int>((pixels + 255) / 256);
    blocks = blocks > 4096 ? 4096 : blocks;
    hsv_full_kernel<<<blocks, 256, 0, stream>>>(src, pixels, blue_index, out_u8, out_f32);
    ENDO_LAUNCH_CHECK();
    return 0;
}

extern "C" int endo_point_brightness(const uint8_t* imgs, int frames, int height, int width, const double* points, int n_points,
                                     const double* projections, const double* extrinsics, const float* visibility, const uint8_t* mask, int d,
                                     double sigma_color, double sigma_space, int32_t* valid, double* depth, float* brightness, void* stream_) {
    if (!imgs || !points || !projections || !extrinsics || !visibility || !mask || !valid || !depth || !brightness) return ENDO_E_BADARG;
    if (frames <= 0 || height <= 0 || width <= 0 || n_points <= 0 || d < 1 || sigma_color <= 0.0 || sigma_space <= 0.0) return ENDO_E_BADARG;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ProfScope prof(kProfSmall, stream, 0.0, 0.0);
    point_brightness_kernel<<<dim3((n_points + 127) / 128, frames), 128, 0, stream>>>(imgs, frames, height, width, points, n_points, projections,
                                                                                    extrinsics, visibility, mask, d / 2, -0.5 / (sigma_color * sigma_color),
                                                                                    -0.5 / (sigma_space * sigma_space), valid, depth, brightness);
    ENDO_LAUNCH_CHECK();
    return 0;
}
