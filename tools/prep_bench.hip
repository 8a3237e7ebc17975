// Streaming shapes for prep_dy (G = d + P x + Q in place over 12 planes of 16 samples, sum of G per channel): which block / grid shape
// reaches the HBM rate of a float4 copy (6.3 TB/s, MI355X_MICROARCH.md).  Development tool, not part of the product; the loops below
// restate the streaming part of net.hip's prep_dy_kernel only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/prep_bench.hip -o tools/bin/prep_bench ;  tools/bin/prep_bench [n] [h] [w] [channels] [maps per sample] [pad floats between samples]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void finish(float part, float* bias, int c) {
    __shared__ float s[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    part = wave_sum(part);
    if (lane == 0) s[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s[w]; atomicAdd(bias + c, t); }
}

// A: the product's shape -- grid (chunks, channels, samples), block-strided float4s, two iterations in flight
template <int NOFIN, int LIN>
__global__ void __launch_bounds__(256) shape_a(float* d, const float* x, int64_t ns, int plane, const float* P, const float* Q, float* bias, int bxn, int chn) {
    const int bxi = LIN ? blockIdx.x % bxn : blockIdx.x;
    const int c = LIN ? (blockIdx.x / bxn) % chn : blockIdx.y;
    const int z = LIN ? blockIdx.x / (bxn * chn) : blockIdx.z;
    const float pc = P[c], qc = Q[c];
    const int64_t base = z * ns + (int64_t)c * plane;
    float part = 0.f;
    const int stride = bxn * blockDim.x * 4;
    int i = (bxi * blockDim.x + threadIdx.x) * 4;
    for (; i + stride < plane; i += 2 * stride) {
        f32x4 g0 = *(const f32x4*)(d + base + i), g1 = *(const f32x4*)(d + base + i + stride);
        const f32x4 x0 = *(const f32x4*)(x + base + i), x1 = *(const f32x4*)(x + base + i + stride);
        for (int e = 0; e < 4; ++e) { g0[e] += fmaf(pc, x0[e], qc); part += g0[e]; }
        for (int e = 0; e < 4; ++e) { g1[e] += fmaf(pc, x1[e], qc); part += g1[e]; }
        *(f32x4*)(d + base + i) = g0; *(f32x4*)(d + base + i + stride) = g1;
    }
    for (; i < plane; i += stride) {
        f32x4 g = *(const f32x4*)(d + base + i); const f32x4 xv = *(const f32x4*)(x + base + i);
        for (int e = 0; e < 4; ++e) { g[e] += fmaf(pc, xv[e], qc); part += g[e]; }
        *(f32x4*)(d + base + i) = g;
    }
    if (NOFIN) { if (part == 12345.678f) bias[c] = part; } else finish(part, bias, c);
}

// B: a block owns ONE contiguous piece of a plane (U float4s per thread, all loads issued before the first use)
template <int U>
__global__ void __launch_bounds__(256) shape_b(float* d, const float* x, int64_t ns, int plane, const float* P, const float* Q, float* bias) {
    const int c = blockIdx.y;
    const float pc = P[c], qc = Q[c];
    const int64_t base = blockIdx.z * ns + (int64_t)c * plane + (int64_t)blockIdx.x * (256 * 4 * U);
    const int left = plane - blockIdx.x * (256 * 4 * U);
    f32x4 g[U], xv[U];
    float part = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = (u * 256 + threadIdx.x) * 4;
        if (i < left) { g[u] = *(const f32x4*)(d + base + i); xv[u] = *(const f32x4*)(x + base + i); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = (u * 256 + threadIdx.x) * 4;
        if (i < left) {
            for (int e = 0; e < 4; ++e) { g[u][e] += fmaf(pc, xv[u][e], qc); part += g[u][e]; }
            *(f32x4*)(d + base + i) = g[u];
        }
    }
    finish(part, bias, c);
}

// C: persistent blocks, grid-stride over (sample, channel, piece) units of U float4s per thread; the next unit's loads are issued before
// this unit's arithmetic and stores
template <int U>
__global__ void __launch_bounds__(256) shape_c(float* d, const float* x, int64_t ns, int plane, int channels, int samples, const float* P, const float* Q, float* bias) {
    const int pieces = (plane + 256 * 4 * U - 1) / (256 * 4 * U);
    const int units = pieces * channels * samples;
    f32x4 g[U], xv[U], gn[U], xn[U];
    auto where = [&](int unit, int& c, int64_t& base, int& left) {
        const int piece = unit % pieces, rest = unit / pieces;
        c = rest % channels;
        const int n = rest / channels;
        base = n * ns + (int64_t)c * plane + (int64_t)piece * (256 * 4 * U);
        left = plane - piece * (256 * 4 * U);
    };
    auto load = [&](int unit, f32x4 (&gg)[U], f32x4 (&xx)[U]) {
        int c, left; int64_t base;
        where(unit, c, base, left);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = (u * 256 + threadIdx.x) * 4;
            if (i < left) { gg[u] = *(const f32x4*)(d + base + i); xx[u] = *(const f32x4*)(x + base + i); }
        }
    };
    int unit = blockIdx.x;
    if (unit >= units) return;
    load(unit, g, xv);
    for (; unit < units; unit += gridDim.x) {
        const int next = unit + gridDim.x;
        if (next < units) load(next, gn, xn);
        int c, left; int64_t base;
        where(unit, c, base, left);
        const float pc = P[c], qc = Q[c];
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = (u * 256 + threadIdx.x) * 4;
            if (i < left) {
                for (int e = 0; e < 4; ++e) { g[u][e] += fmaf(pc, xv[u][e], qc); part += g[u][e]; }
                *(f32x4*)(d + base + i) = g[u];
            }
        }
        part = wave_sum(part);
        if ((threadIdx.x & 63) == 0) atomicAdd(bias + c, part);
#pragma unroll
        for (int u = 0; u < U; ++u) { g[u] = gn[u]; xv[u] = xn[u]; }
    }
}

// reference: the same bytes as a plain copy-like pass (read 2, write 1) with no per-channel structure
__global__ void __launch_bounds__(256) flat(float* d, const float* x, int64_t total4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 g = ((const f32x4*)d)[i]; const f32x4 xv = ((const f32x4*)x)[i];
        for (int e = 0; e < 4; ++e) g[e] += 0.5f * xv[e];
        ((f32x4*)d)[i] = g;
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16, h = argc > 2 ? atoi(argv[2]) : 256, w = argc > 3 ? atoi(argv[3]) : 320, ch = argc > 4 ? atoi(argv[4]) : 12;
    const int t = argc > 5 ? atoi(argv[5]) : 96, plane = h * w;
    const int64_t pad = argc > 6 ? atoll(argv[6]) : 0;          // extra floats between samples
    const int64_t ns = (int64_t)t * plane + pad;
    float *d, *x, *P, *Q, *bias;
    CK(hipMalloc(&d, n * ns * 4)); CK(hipMalloc(&x, n * ns * 4)); CK(hipMalloc(&P, 4 * t)); CK(hipMalloc(&Q, 4 * t)); CK(hipMalloc(&bias, 4 * t));
    CK(hipMemset(d, 0, n * ns * 4)); CK(hipMemset(x, 0, n * ns * 4)); CK(hipMemset(P, 0, 4 * t)); CK(hipMemset(Q, 0, 4 * t)); CK(hipMemset(bias, 0, 4 * t));
    const double bytes = 12.0 * n * plane * ch;
    printf("prep_dy shapes: %d samples of %d x %d, %d channels of a %d-map buffer, sample stride %lld floats (pad %lld): %.0f MB per launch\n", n, h, w, ch, t, (long long)ns, (long long)pad, bytes / 1e6);
    struct V { std::string name; std::function<void()> run; };
    std::vector<V> vs;
    int bx = (plane + 4095) / 4096; bx = bx < 1 ? 1 : (bx > 32 ? 32 : bx);
    vs.push_back({"A: product shape (strided, 2 in flight)", [&] { shape_a<0, 0><<<dim3(bx, ch, n), 256>>>(d, x, ns, plane, P, Q, bias, bx, ch); }});
    vs.push_back({"A without the block reduction and the atomic", [&] { shape_a<1, 0><<<dim3(bx, ch, n), 256>>>(d, x, ns, plane, P, Q, bias, bx, ch); }});
    vs.push_back({"A on a 1-D grid", [&] { shape_a<0, 1><<<dim3(bx * ch * n), 256>>>(d, x, ns, plane, P, Q, bias, bx, ch); }});
    vs.push_back({"A on a 1-D grid, no reduction / atomic", [&] { shape_a<1, 1><<<dim3(bx * ch * n), 256>>>(d, x, ns, plane, P, Q, bias, bx, ch); }});
    vs.push_back({"B: contiguous piece per block, 4 float4 / thread", [&] { shape_b<4><<<dim3((plane + 4095) / 4096, ch, n), 256>>>(d, x, ns, plane, P, Q, bias); }});
    vs.push_back({"B: contiguous piece per block, 8 float4 / thread", [&] { shape_b<8><<<dim3((plane + 8191) / 8192, ch, n), 256>>>(d, x, ns, plane, P, Q, bias); }});
    vs.push_back({"B: contiguous piece per block, 2 float4 / thread", [&] { shape_b<2><<<dim3((plane + 2047) / 2048, ch, n), 256>>>(d, x, ns, plane, P, Q, bias); }});
    vs.push_back({"flat read-2-write-1 over the same bytes (contiguous)", [&] { flat<<<256 * 8, 256>>>(d, x, (int64_t)n * plane * ch / 4); }});
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto& v : vs) {
        for (int i = 0; i < 3; ++i) v.run();
        CK(hipDeviceSynchronize());
        const int reps = 30;
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) v.run();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-58s %7.1f us  %5.2f TB/s\n", v.name.c_str(), ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    }
    return 0;
}
