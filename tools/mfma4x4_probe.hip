// Probe of v_mfma_f32_4x4x1_16b_f32 operand / result layout and issue rate on gfx950 (development tool).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_probe.hip -o tools/bin/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* out) {
    const int l = threadIdx.x;
    // a encodes (block, i) = (l / 4, l % 4) as 100*block + 10*i + 1 ; b encodes (block, j) as 1000 * (j + 1) (same for all blocks -> look at products)
    const float a = 100.f * (l / 4) + 10.f * (l % 4) + 1.f;
    const float b = (l % 4 == 0) ? 1.f : (l % 4 == 1 ? 0.001f : (l % 4 == 2 ? 1e-6f : 1e-9f));
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

__global__ void rate(float* out, int iters) {
    const int l = threadIdx.x;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    float a = l * 0.001f, b = 1.0f + l * 1e-4f;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c7, 0, 0, 0);
    }
    f32x4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    out[blockIdx.x * blockDim.x + l] = s[0] + s[1] + s[2] + s[3];
}

__global__ void rate16(float* out, int iters) {
    const int l = threadIdx.x;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a = l * 0.001f, b = 1.0f + l * 1e-4f;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    f32x4 s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + l] = s[0] + s[1] + s[2] + s[3];
}

int main() {
    float* d; hipMalloc(&d, 1 << 22);
    probe<<<1, 64>>>(d);
    std::vector<float> h(256);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) {
        if (l < 8 || l % 16 == 5) printf("lane %2d: %14.9f %14.9f %14.9f %14.9f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves = 1; waves <= 2; ++waves) {
        rate<<<1024, 256 * waves>>>(d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0); rate<<<1024, 256 * waves>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = 1024.0 * 4 * waves * iters * 8 * 512.0;
        printf("4x4x1 : %d waves/SIMD-ish: %.2f ms  %.1f TFLOP/s\n", waves, ms, flop / ms / 1e9);
        hipEventRecord(e0); rate16<<<1024, 256 * waves>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        flop = 1024.0 * 4 * waves * iters * 4 * 2048.0;
        printf("16x16x4: %d waves/SIMD-ish: %.2f ms  %.1f TFLOP/s\n", waves, ms, flop / ms / 1e9);
    }
    return 0;
}
