// Issue rate of the matrix instructions the kernels use or could use, and of packed fp32 FMA, on gfx950 (development tool):
// cycles per instruction per SIMD with 1 and 2 waves per SIMD, and whether MFMA and VALU work of two waves overlaps.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip -o tools/bin/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void rate(float* out, int iters, long long* cycles) {
    const int l = threadIdx.x;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a = l * 0.001f, b = 1.0f + l * 1e-4f;
    s16x4 ab = {1, 2, 3, 4}, bb = {5, 6, 7, 8};
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(0.001f * (l + i)); b8[i] = (__bf16)(1.0f + 0.01f * i); }
    f32x2 v0 = {a, b}, v1 = {b, a}, v2 = {a, a}, v3 = {b, b};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        } else if constexpr (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, bb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, bb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, bb, c3, 0, 0, 0);
        } else if constexpr (KIND == 2) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c3, 0, 0, 0);
        } else if constexpr (KIND == 3) {          // packed fp32 FMA: 2 FMAs per lane and instruction
            v0 = __builtin_elementwise_fma(v0, v1, v2); v1 = __builtin_elementwise_fma(v1, v2, v3);
            v2 = __builtin_elementwise_fma(v2, v3, v0); v3 = __builtin_elementwise_fma(v3, v0, v1);
        } else {          // fp32 MFMA and packed FMA interleaved in ONE wave
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); v0 = __builtin_elementwise_fma(v0, v1, v2); v1 = __builtin_elementwise_fma(v1, v2, v3);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0); v2 = __builtin_elementwise_fma(v2, v3, v0); v3 = __builtin_elementwise_fma(v3, v0, v1);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); v0 = __builtin_elementwise_fma(v0, v1, v2); v1 = __builtin_elementwise_fma(v1, v2, v3);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0); v2 = __builtin_elementwise_fma(v2, v3, v0); v3 = __builtin_elementwise_fma(v3, v0, v1);
        }
    }
    const long long t1 = clock64();
    f32x4 s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + l] = s[0] + s[1] + s[2] + s[3] + v0[0] + v1[1] + v2[0] + v3[1];
    if (blockIdx.x == 0 && l == 0) *cycles = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_iter, float* d, long long* dc) {
    const int iters = 20000;
    for (int waves = 1; waves <= 8; waves *= 2) {          // waves per CU: 4 = one per SIMD, 8 = two per SIMD
        rate<KIND><<<256, 64 * waves>>>(d, iters, dc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        rate<KIND><<<256, 64 * waves>>>(d, iters, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
        printf("%-44s waves/CU %d: %7.2f us, %6.1f shader clocks per instruction and wave (s_memtime), %6.2f ns per instruction\n", name, waves,
               ms * 1e3, (double)cyc / (iters * per_iter), ms * 1e6 / (iters * per_iter));
    }
}

int main() {
    float* d; hipMalloc(&d, 1 << 22);
    long long* dc; hipMalloc(&dc, 8);
    run<0>("v_mfma_f32_16x16x4_f32", 4, d, dc);
    run<1>("v_mfma_f32_16x16x16_bf16", 4, d, dc);
    run<2>("v_mfma_f32_16x16x32_bf16", 4, d, dc);
    run<3>("v_pk_fma_f32", 4, d, dc);
    run<4>("1 fp32 MFMA + 2 v_pk_fma_f32 interleaved (per triple)", 4, d, dc);
    return 0;
}
