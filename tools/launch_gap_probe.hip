// Launch-gap probe (development tool): N dependent launches in one stream of (a) the same kernel, (b) two instantiations alternating,
// (c) the same kernel with a tiny other kernel in between; kernel duration ~ spin iterations.  Reports (total - N * kernel) / N.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_gap_probe.hip -o tools/bin/launch_gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int V>
__global__ void __launch_bounds__(256) work(float* p, int iters, int lds_dummy) {
    extern __shared__ float sm[];
    float v = p[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
    if (lds_dummy < 0) sm[threadIdx.x] = v;
    p[blockIdx.x * 256 + threadIdx.x] = v + V;
}
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }

int main() {
    float* d; CK(hipMalloc(&d, 4 << 20)); CK(hipMemset(d, 0, 4 << 20));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int N = 400;
    for (int blocks : {512, 2048}) for (int iters : {2000, 20000}) for (size_t lds : {(size_t)0, (size_t)48 * 1024}) {
        if (lds) { CK(hipFuncSetAttribute((const void*)work<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); CK(hipFuncSetAttribute((const void*)work<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); }
        float ms_one, ms[3];
        // single kernel duration
        work<0><<<blocks, 256, lds>>>(d, iters, 0); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a)); work<0><<<blocks, 256, lds>>>(d, iters, 0); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms_one, a, b));
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            for (int i = 0; i < N; ++i) {
                if (mode == 1 && (i & 1)) work<1><<<blocks, 256, lds>>>(d, iters, 0);
                else work<0><<<blocks, 256, lds>>>(d, iters, 0);
                if (mode == 2) tiny<<<1, 64>>>(d);
            }
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms[mode], a, b));
        }
        printf("blocks %4d iters %5d lds %5zu: one launch %.1f us | per launch: same kernel %.2f us, alternating instantiations %.2f us, same + tiny kernel between %.2f us\n",
               blocks, iters, lds, ms_one * 1e3, ms[0] * 1e3 / N, ms[1] * 1e3 / N, ms[2] * 1e3 / N);
    }
    return 0;
}
