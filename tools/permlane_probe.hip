// v_permlane16_swap / v_permlane32_swap semantics probe (development tool): prints what the two results of the builtins hold per 16-lane row.
//   hipcc --offload-arch=gfx950 -O3 tools/permlane_probe.hip -o tools/bin/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned u = threadIdx.x;                       // lane id
    const unsigned w = 1000 + threadIdx.x;
    const auto a = __builtin_amdgcn_permlane16_swap(u, w, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(u, w, false, false);
    const auto c = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    out[threadIdx.x] = a[0]; out[64 + threadIdx.x] = a[1]; out[128 + threadIdx.x] = b[0]; out[192 + threadIdx.x] = b[1];
    out[256 + threadIdx.x] = c[0]; out[320 + threadIdx.x] = c[1]; out[384 + threadIdx.x] = d[0]; out[448 + threadIdx.x] = d[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512 * 4); k<<<1, 64>>>(d); unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"p16(u,w)[0]", "p16(u,w)[1]", "p32(u,w)[0]", "p32(u,w)[1]", "p16(u,u)[0]", "p16(u,u)[1]", "p32(u,u)[0]", "p32(u,u)[1]"};
    for (int r = 0; r < 8; ++r) { printf("%-12s rows start with:", names[r]); for (int row = 0; row < 4; ++row) printf(" %u", h[r * 64 + row * 16]); printf("\n"); }
    return 0;
}
