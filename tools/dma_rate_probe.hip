// LDS-DMA issue-rate probe (development tool): how many cycles of a CU does one global->LDS DMA instruction cost, dword against 16-byte form,
// against ordinary 16-byte loads into registers, with 1 / 2 / 4 / 8 waves per CU issuing?  Sources are a few KB per block (L2 / L1 hits).
//   hipcc --offload-arch=gfx950 -O3 tools/dma_rate_probe.hip -o tools/bin/dma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>          // 0 = dword DMA, 1 = 16-byte DMA, 2 = 16-byte loads into registers
__global__ void __launch_bounds__(512) probe(const float* __restrict__ src, float* __restrict__ out, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = src + (blockIdx.x & 63) * 8192 + wave * 1024;          // 32 KB per block position, reused: cache hits
    float* dst = lds + wave * 1024;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) __builtin_amdgcn_global_load_lds((gptr_t)(base + ((i + k) & 3) * 64 + lane), (lptr_t)(dst + k * 64), 4, 0, 0);
            else if (MODE == 1) __builtin_amdgcn_global_load_lds((gptr_t)(base + ((i + k) & 3) * 256 + 4 * lane), (lptr_t)(dst + (k & 3) * 256), 16, 0, 0);
            else { const f32x4 v = *reinterpret_cast<const f32x4*>(base + ((i + k) & 3) * 256 + 4 * lane); acc += v; }
        }
        if (MODE != 2 && (i & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[threadIdx.x] = acc[0] + lds[threadIdx.x];
}

int main() {
    float* src; CK(hipMalloc(&src, 64 * 8192 * 4)); CK(hipMemset(src, 0, 64 * 8192 * 4));
    float* out; CK(hipMalloc(&out, 4096));
    unsigned long long* cyc; CK(hipMalloc(&cyc, 8));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 2000;
    const char* names[3] = {"dword LDS-DMA", "16-byte LDS-DMA", "16-byte load to registers"};
    for (int mode = 0; mode < 3; ++mode)
        for (int waves : {1, 2, 4, 8}) {
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(a));
                if (mode == 0) probe<0><<<cus, 64 * waves, 32768>>>(src, out, iters, cyc);
                else if (mode == 1) probe<1><<<cus, 64 * waves, 32768>>>(src, out, iters, cyc);
                else probe<2><<<cus, 64 * waves, 32768>>>(src, out, iters, cyc);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            }
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            const double instrs = (double)iters * 8 * waves;          // per CU
            const double bytes = instrs * 64 * (mode == 0 ? 4 : 16);
            printf("%-28s %d wave(s) per CU: %7.1f cycles of the block per instruction-of-any-wave (%.1f per wave-instruction), %6.1f B/clk/CU, %7.1f us, %.2f TB/s over %d CUs\n",
                   names[mode], waves, (double)c / instrs, (double)c / (iters * 8.0), bytes / (double)c, ms * 1e3, bytes * cus / (ms * 1e-3) / 1e12, cus);
        }
    return 0;
}
