// New-channel data-gradient passes of a dense block stand-alone: the per-tile kernel (dgrad_block_kernel with count = 12) with its
// diagnostic EXP masks -- where the time of these short-loop kernels goes -- and the persistent form that replaced it in round 5
// (dgrad_newmap_kernel).  Development tool, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/nl_bench.hip -o tools/bin/nl_bench
//   tools/bin/nl_bench [n] [h] [w]
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include <functional>

#include "../endoscopydepthestimation-pytorch_amd/csrc/dgrad_kernels.h"
#include "../endoscopydepthestimation-pytorch_amd/csrc/wgrad_taps_kernels.h"
#include "../endoscopydepthestimation-pytorch_amd/csrc/wgrad_nsplit_kernels.h"
#include "../endoscopydepthestimation-pytorch_amd/csrc/dgrad_block_kernels.h"
#include "../endoscopydepthestimation-pytorch_amd/csrc/dgrad_newmap_kernels.h"

using namespace endo;

// stubs for the profiling hooks referenced by common.h
endo::ProfScope::ProfScope(int f, hipStream_t s, double, double) : family(f), stream(s), slot(nullptr) {}
endo::ProfScope::~ProfScope() {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static float* dev_random(size_t n, float lo, float hi, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = lo + (hi - lo) * ((s >> 8) & 0xFFFF) / 65535.0f; }
    float* d; CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

struct Variant { std::string name; std::function<int(hipStream_t)> run; };

static void bench(std::vector<Variant>& vs, float* out, size_t out_n, double flops) {
    std::vector<float> ref, cur(out_n);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto& v : vs) {
        CK(hipMemset(out, 0, out_n * sizeof(float)));
        int rc = v.run(0);
        if (rc) { printf("%-40s launch failed rc=%d\n", v.name.c_str(), rc); continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(cur.data(), out, out_n * sizeof(float), hipMemcpyDeviceToHost));
        double maxdiff = 0, maxref = 0;
        if (ref.empty()) ref = cur;
        size_t bad = 0;
        for (size_t i = 0; i < out_n; ++i) {
            const double d = fabs((double)cur[i] - ref[i]);
            if (d > 1e-3 && v.name.find("persistent") != std::string::npos && bad++ < 12) printf("   mismatch at %zu: %g vs %g\n", i, cur[i], ref[i]);
            maxdiff = fmax(maxdiff, d); maxref = fmax(maxref, fabs((double)ref[i]));
        }
        if (bad) printf("   %zu mismatches of %zu\n", bad, out_n);
        for (int i = 0; i < 3; ++i) v.run(0);
        CK(hipDeviceSynchronize());
        const int reps = 20;
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) v.run(0);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-44s %8.1f us  %6.1f TFLOP/s   max|diff| %.2e (max|ref| %.2e)\n", v.name.c_str(), ms / reps * 1e3, flops / (ms / reps * 1e-3) / 1e12, maxdiff, maxref);
    }
}


template <int NL, int EXP, int VEC>
static int run_nl(const DgradBlockParams& p, hipStream_t s) { return launch_dgrad_block<NL, 2, 3, 1, EXP, 1, VEC>(p, s); }

template <int NL>
static void nl_section(DgradBlockParams p, int n, int64_t plane) {
    const double flops = 2.0 * n * plane * 12 * 12 * 9 * NL;
    printf("---- NL = %d (MFMA floor at 157 TF with 16-wide groups: %.0f us; algorithmic HBM %.0f MB) ----\n", NL,
           2.0 * n * plane * 16 * 12 * 9 * NL / 157.3e12 * 1e6, 4.0 * n * plane * (12.0 * NL + 36) / 1e6);
    std::vector<Variant> vs;
    vs.push_back({"library (16-byte dY DMA)", [&](hipStream_t s) { return run_nl<NL, 0, 4>(p, s); }});
    DgradBlockParams one = p;
    one.slot_stride = 0;
    vs.push_back({"BN sums in ONE copy (same-address atomics)", [&](hipStream_t s) { return run_nl<NL, 0, 4>(one, s); }});
    vs.push_back({"persistent blocks (dgrad_newmap_kernel)", [&](hipStream_t s) { return launch_dgrad_newmap<NL>(p, s); }});
    vs.push_back({"dword dY DMA", [&](hipStream_t s) { return run_nl<NL, 0, 1>(p, s); }});
    vs.push_back({"no x / dbuf loads (1)", [&](hipStream_t s) { return run_nl<NL, 1, 4>(p, s); }});
    vs.push_back({"no stores (2)", [&](hipStream_t s) { return run_nl<NL, 2, 4>(p, s); }});
    vs.push_back({"no loads, no stores (3)", [&](hipStream_t s) { return run_nl<NL, 3, 4>(p, s); }});
    vs.push_back({"no BN-sum reduction (8)", [&](hipStream_t s) { return run_nl<NL, 8, 4>(p, s); }});
    vs.push_back({"no dY tile load (16)", [&](hipStream_t s) { return run_nl<NL, 16, 4>(p, s); }});
    vs.push_back({"trivial epilogue (32+8)", [&](hipStream_t s) { return run_nl<NL, 40, 4>(p, s); }});
    vs.push_back({"trivial epi, no ld/st (43)", [&](hipStream_t s) { return run_nl<NL, 43, 4>(p, s); }});
    vs.push_back({"+ weights once (47)", [&](hipStream_t s) { return run_nl<NL, 47, 4>(p, s); }});
    vs.push_back({"+ no dY tile load (63): MFMA + LDS reads", [&](hipStream_t s) { return run_nl<NL, 63, 4>(p, s); }});
    bench(vs, p.out, (size_t)12 * plane, flops);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 16;
    const int h = argc > 2 ? atoi(argv[2]) : 256;
    const int w = argc > 3 ? atoi(argv[3]) : 320;
    const int t = 96;
    const int64_t plane = (int64_t)h * w;
    printf("new-channel passes: N=%d %dx%d, level buffer of %d maps\n", n, h, w, t);
    float* buf = dev_random((size_t)n * t * plane, -1.f, 1.f, 1);
    float* gbuf = dev_random((size_t)n * t * plane, -1.f, 1.f, 2);
    const int cin = 84;
    float* wgt = dev_random((size_t)12 * cin * 9, -0.05f, 0.05f, 3);
    float* gamma = dev_random(cin, 0.8f, 1.2f, 5);
    float* beta = dev_random(cin, -0.1f, 0.1f, 6);
    float* saved; CK(hipMalloc(&saved, 2 * cin * sizeof(float)));
    std::vector<float> hs(2 * cin); for (int c = 0; c < cin; ++c) { hs[2 * c] = 0.01f * (c % 7); hs[2 * c + 1] = 1.7f; }
    CK(hipMemcpy(saved, hs.data(), hs.size() * sizeof(float), hipMemcpyHostToDevice));
    double* scratch; CK(hipMalloc(&scratch, kBnSlots * 2 * t * sizeof(double))); CK(hipMemset(scratch, 0, kBnSlots * 2 * t * sizeof(double)));
    DgradBlockParams p{};
    p.slot_stride = 2 * t;
    p.n = n; p.h = h; p.w = w;
    p.g = gbuf + 60 * plane; p.g_ns = t * plane; p.g_cs = (int)plane; p.g_w = w;
    p.x = buf + 48 * plane; p.out = gbuf + 48 * plane; p.ns = t * plane; p.cs = (int)plane; p.count = 12; p.acc_from = 0; p.w_ci_off = 48;
    for (int j = 0; j < 4; ++j) {
        p.wgt[j] = wgt; p.w_cin[j] = cin; p.saved[j] = saved + 96; p.gamma[j] = gamma + 48; p.beta[j] = beta + 48; p.scratch[j] = scratch + 96;
    }
    nl_section<1>(p, n, plane);
    nl_section<2>(p, n, plane);
    nl_section<3>(p, n, plane);
    return 0;
}
