// Issue cost of the vector instructions the bf16x3 split uses (development probe): each kernel runs ITER iterations of 16 independent
// instructions of one kind per wave, 4 waves per CU (one per SIMD) or 8 (two per SIMD); cycles per instruction = time * clock / count.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o tools/bin/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 4096;

#define BODY16(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) INS(14) INS(15)

template <int KIND>
__global__ void __launch_bounds__(256) probe(float* out) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 0.001f + i;
    typedef float f2 __attribute__((ext_vector_type(2)));
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(v[i + 16]));
            if (KIND == 1) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(v[i]) : "v"(v[i]));
            if (KIND == 2) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(v[i + 16]));
            if (KIND == 3) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(v[i + 16]), "v"(v[(i + 1) & 15]));
            if (KIND == 4) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(v[i]) : "v"(v[i]));
            if (KIND == 5) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(v[i + 16]), "v"(v[(i + 1) & 15]));
            if (KIND == 6) asm volatile("v_max_f32 %0, 0, %1" : "=v"(v[i]) : "v"(v[i]));
            if (KIND == 7) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "=v"(v[i]) : "v"(v[i]));
        }
        if (KIND == 8) {          // v_pk_add_f32 on 8 register pairs
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f2 a = {v[i], v[i + 1]}, b = {v[i + 16], v[i + 17]};
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(b));
                v[i] = a[0]; v[i + 1] = a[1];
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
int run(const char* name, int per_iter, float* out) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
        probe<KIND><<<256 * blocks_per_cu, 256>>>(out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, 0));
        probe<KIND><<<256 * blocks_per_cu, 256>>>(out);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double cycles = ms * 1e-3 * 2.4e9;          // nominal clock
        printf("%-28s %d wave(s) per SIMD: %7.1f us -> %.2f cycles per instruction and wave at 2.4 GHz\n", name, blocks_per_cu, ms * 1e3, cycles / (double(ITER) * per_iter));
    }
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 512 * 256 * 4));
    run<0>("v_cvt_pk_bf16_f32", 16, out);
    run<1>("v_and_b32", 16, out);
    run<2>("v_sub_f32", 16, out);
    run<3>("v_perm_b32", 16, out);
    run<4>("v_lshlrev_b32", 16, out);
    run<5>("v_fma_f32", 16, out);
    run<6>("v_max_f32", 16, out);
    run<7>("v_cndmask_b32", 16, out);
    run<8>("v_pk_add_f32", 8, out);
    return 0;
}
