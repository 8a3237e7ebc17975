// What a new-map data-gradient pass of the bf16-storage family is made of (development tool, not part of the product): times
// bf16_conv_kernel<3, 3, kEpiDgradBn, 4, 1, 0, 0, 8> -- the per-layer data gradient with respect to a dense block's new maps -- with parts of it
// switched off (tools/make_conv_diag.py adds the masks to a copy of the kernel header).
//   python tools/make_conv_diag.py
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/bf16_dgrad_variants.hip -o tools/bin/bf16_dgrad_variants
//   tools/bin/bf16_dgrad_variants [j] [n] [h] [w]        (j = 1..3: the pass writes 12 j channels)
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <string>
#include <functional>

#include "bin/bf16_conv_diag_kernels.h"

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
    const int j = argc > 1 ? atoi(argv[1]) : 3;
    const int n = argc > 2 ? atoi(argv[2]) : 16, h = argc > 3 ? atoi(argv[3]) : 256, w = argc > 4 ? atoi(argv[4]) : 320;
    const int t = 192, c0 = 48, blk = 32;
    const size_t px = static_cast<size_t>(n) * h * w;
    uint16_t* act = dev_random_bf16(px * t, -1.f, 1.f, 1);
    uint16_t* dbuf = dev_random_bf16(px * t, -0.01f, 0.01f, 3);
    const int wgroups = 4;
    uint16_t* wgt = dev_random_bf16(static_cast<size_t>(wgroups) * 9 * 3 * 16 * 32, -0.1f, 0.1f, 2);
    std::vector<float> hs(2 * 256), hg(256), hb(256);
    for (int c = 0; c < 256; ++c) { hs[2 * c] = 0.01f * (c % 7) - 0.02f; hs[2 * c + 1] = 1.5f + 0.01f * (c % 5); hg[c] = 0.8f + 0.01f * (c % 11); hb[c] = 0.05f * (c % 3) - 0.05f; }
    float *saved, *gamma, *beta;
    CK(hipMalloc(&saved, hs.size() * 4)); CK(hipMemcpy(saved, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&gamma, hg.size() * 4)); CK(hipMemcpy(gamma, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&beta, hb.size() * 4)); CK(hipMemcpy(beta, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    double* sums; CK(hipMalloc(&sums, 2 * 256 * sizeof(double)));

    Conv16Params p{};
    p.n = n; p.h = h; p.w = w;
    p.in = dbuf; p.in_t = t; p.in_blk = blk; p.in_h = h; p.in_w = w; p.in_ns = static_cast<int64_t>(h) * w * t; p.ic0 = c0 + 12 * j; p.cin = 12;
    p.wgt = wgt;
    p.out = dbuf; p.out_t = t; p.out_blk = blk; p.out_ns = p.in_ns; p.oc0 = c0; p.cout = 12 * j;
    p.x = act; p.x_saved = saved; p.gamma = gamma; p.beta = beta;
    p.out_sums = sums; p.co_off = c0; p.grp0 = c0 / 48; p.wgroups = wgroups; p.sr_salt = 12345u;

    std::vector<Variant> vs;
#define V(name, exp) vs.push_back({name, [&](hipStream_t s) { return launch_bf16_conv<3, 3, kEpiDgradBn, 4, 1, exp, 0, 8>(p, s); }})
    vs.push_back({"8-wave blocks over 16-row tiles (until r03_ad)", [&](hipStream_t s) { return launch_bf16_conv<3, 3, kEpiDgradBn, 8, 2, 0>(p, s); }});
    V("product", 0);
    V("no matrix phase", 1);
    V("no input-tile loads", 2);
    V("no epilogue loads", 8);
    V("no stores", 16);
    V("no BatchNorm-backward sums", 32);
    V("round to nearest", 64);
    V("no epilogue loads, no stores", 24);
    V("no tile loads, no epilogue loads, no stores", 26);
    V("only the tile loads + matrix phase", 8 | 16 | 32);
    V("only the epilogue (loads, arithmetic, stores, sums)", 1 | 2 | 4);
    V("nothing but the prologue", 1 | 2 | 4 | 8 | 16 | 32);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double bytes = static_cast<double>(px) * (2.0 * 12 + 6.0 * 12 * j);
    printf("bf16 new-map data gradient  %d x %d x %d  12 -> %d channels   algorithmic %.1f MB\n", n, h, w, 12 * j, bytes / 1e6);
    for (auto& v : vs) {
        CK(hipMemset(sums, 0, 2 * 256 * sizeof(double)));
        int rc = v.run(0);
        if (rc) { printf("%-52s launch failed rc=%d\n", v.name.c_str(), rc); continue; }
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("%-52s FAILED: %s\n", v.name.c_str(), hipGetErrorString(e)); return 1; }
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
        printf("%-52s %8.1f us   %5.2f TB/s algorithmic\n", v.name.c_str(), best * 1e3, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
