// bf16 NHWC convolution microbenchmark (development tool, not part of the product): times bf16_conv_kernel in its block-shape /
// prefetch-depth / register-cap variants on the dense layers of the benchmark shape and cross-checks their outputs bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/bf16_conv_variants.hip -o tools/bin/bf16_conv_variants
//   tools/bin/bf16_conv_variants [cin] [n] [h] [w] [blk]     (blk = channels per block of the buffer, 0 = plain NHWC)
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <string>
#include <functional>

#include "../endoscopydepthestimation-pytorch_amd/csrc/bf16_conv_kernels.h"

using namespace endo;

endo::ProfScope::ProfScope(int f, hipStream_t s, double, double) : family(f), stream(s), slot(nullptr) {}
endo::ProfScope::~ProfScope() {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t* dev_random_bf16(size_t n, float lo, float hi, unsigned seed) {
    std::vector<uint16_t> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float v = lo + (hi - lo) * ((s >> 8) & 0xFFFF) / 65535.0f;
        unsigned bits; memcpy(&bits, &v, 4);
        h[i] = static_cast<uint16_t>((bits + 0x7fffu + ((bits >> 16) & 1u)) >> 16);
    }
    uint16_t* d; CK(hipMalloc(&d, n * 2));
    CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}

struct Variant { std::string name; std::function<int(hipStream_t)> run; };

int main(int argc, char** argv) {
    const int cin = argc > 1 ? atoi(argv[1]) : 180;
    const int n = argc > 2 ? atoi(argv[2]) : 16, h = argc > 3 ? atoi(argv[3]) : 256, w = argc > 4 ? atoi(argv[4]) : 320;
    const int blk = argc > 5 ? atoi(argv[5]) : 0;
    const int cout = 12, t = cin + cout > 192 ? cin + cout : 192;
    const size_t px = static_cast<size_t>(n) * h * w;
    uint16_t* buf = dev_random_bf16(px * t, -1.f, 1.f, 1);
    std::vector<float> hbn(2 * 512);
    for (int c = 0; c < 512; ++c) { hbn[2 * c] = 0.5f + 0.01f * c; hbn[2 * c + 1] = 0.1f - 0.002f * c; }
    float* bn; CK(hipMalloc(&bn, hbn.size() * 4)); CK(hipMemcpy(bn, hbn.data(), hbn.size() * 4, hipMemcpyHostToDevice));
    const int nchunks = (cin + 31) / 32;
    uint16_t* wgt = dev_random_bf16(static_cast<size_t>(nchunks) * 9 * 16 * 32, -0.1f, 0.1f, 2);
    double* sums; CK(hipMalloc(&sums, 2 * 16 * sizeof(double)));

    Conv16Params p{};
    p.n = n; p.h = h; p.w = w; p.in = buf; p.in_ns = static_cast<int64_t>(h) * w * t; p.in_t = t; p.in_h = h; p.in_w = w;
    p.ic0 = 0; p.cin = cin; p.bn = bn; p.wgt = wgt; p.out = buf; p.out_ns = p.in_ns; p.out_t = t; p.oc0 = cin; p.cout = cout; p.out_sums = sums; p.in_blk = blk; p.out_blk = blk;

    std::vector<Variant> vs;
    vs.push_back({"waves 4, 2 waves/SIMD", [&](hipStream_t s) { return launch_bf16_conv<3, 1, 0, 4, 2>(p, s); }});
    vs.push_back({"waves 8, 4 waves/SIMD", [&](hipStream_t s) { return launch_bf16_conv<3, 1, 0, 8, 4>(p, s); }});
    vs.push_back({"waves 8, 2 waves/SIMD (1 block/CU)", [&](hipStream_t s) { return launch_bf16_conv<3, 1, 0, 8, 2>(p, s); }});
    vs.push_back({"waves 4, 3 waves/SIMD", [&](hipStream_t s) { return launch_bf16_conv<3, 1, 0, 4, 3>(p, s); }});
    static Conv16Params q = p, r = p;
    q.out_sums = nullptr;
    r.bn = nullptr;
    vs.push_back({"waves 8, 4 waves/SIMD, no statistics", [&](hipStream_t s) { return launch_bf16_conv<3, 1, 0, 8, 4>(q, s); }});
    vs.push_back({"waves 8, 4 waves/SIMD, raw input", [&](hipStream_t s) { return launch_bf16_conv<3, 1, 0, 8, 4>(r, s); }});
    vs.push_back({"waves 8, 4 w/SIMD, no matrix phase", [&](hipStream_t s) { return launch_bf16_conv<3, 1, 0, 8, 4, 1>(p, s); }});
    vs.push_back({"waves 8, 4 w/SIMD, no activation loads", [&](hipStream_t s) { return launch_bf16_conv<3, 1, 0, 8, 4, 2>(p, s); }});
    vs.push_back({"waves 8, 4 w/SIMD, no BN / stage writes", [&](hipStream_t s) { return launch_bf16_conv<3, 1, 0, 8, 4, 4>(p, s); }});
    vs.push_back({"waves 8, 4 w/SIMD, loads only", [&](hipStream_t s) { return launch_bf16_conv<3, 1, 0, 8, 4, 5>(p, s); }});
    vs.push_back({"waves 8, 4 w/SIMD, matrix phase only", [&](hipStream_t s) { return launch_bf16_conv<3, 1, 0, 8, 4, 6>(p, s); }});
    const size_t out_bytes = px * t * 2;
    std::vector<uint16_t> ref, cur(px * t);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double bytes = static_cast<double>(px) * (cin + cout) * 2;
    printf("bf16 3x3 conv  %d x %d x %d  cin %d -> %d  channel block %d   algorithmic %.1f MB\n", n, h, w, cin, cout, blk ? blk : t, bytes / 1e6);
    for (auto& v : vs) {
        CK(hipMemset(sums, 0, 2 * 16 * sizeof(double)));
        int rc = v.run(0);
        if (rc) { printf("%-44s launch failed rc=%d\n", v.name.c_str(), rc); continue; }
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("%-44s FAILED: %s\n", v.name.c_str(), hipGetErrorString(e)); return 1; }
        CK(hipMemcpy(cur.data(), buf, out_bytes, hipMemcpyDeviceToHost));
        size_t diff = 0;
        if (ref.empty()) ref = cur;
        for (size_t i = 0; i < cur.size(); ++i) diff += cur[i] != ref[i];
        for (int i = 0; i < 3; ++i) v.run(0);
        CK(hipDeviceSynchronize());
        const int reps = 20;
        float best = 1e30f;
        for (int rr = 0; rr < 3; ++rr) {
            CK(hipEventRecord(a, 0));
            for (int i = 0; i < reps; ++i) v.run(0);
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            best = fminf(best, ms / reps);
        }
        printf("%-44s %8.1f us   %5.2f TB/s algorithmic   differing values %zu\n", v.name.c_str(), best * 1e3, bytes / (best * 1e-3) / 1e12, diff);
    }
    return 0;
}
