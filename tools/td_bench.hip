// Transition-down data-gradient microbenchmark (development tool): conv_dma_kernel<1, 16, 2, IN_UNPOOL, EPI_DGRAD_BN> (one block per tile and
// 32 output channels) against td_dgrad_kernel (persistent blocks, td_dgrad_kernels.h); cross-checks outputs and BN-backward sums.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/td_bench.hip -o tools/bin/td_bench
//   tools/bin/td_bench [C] [n] [h] [w]      (C = 96 at level 0, 144 at level 1; h, w = the full-resolution side)
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include <functional>
#include "../endoscopydepthestimation-pytorch_amd/csrc/td_dgrad_kernels.h"
using namespace endo;
endo::ProfScope::ProfScope(int f, hipStream_t s, double, double) : family(f), stream(s), slot(nullptr) {}
endo::ProfScope::~ProfScope() {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
static unsigned rs = 12345u;
static float rnd(float lo, float hi) { rs = rs * 1664525u + 1013904223u; return lo + (hi - lo) * ((rs >> 8) & 0xFFFF) / 65535.0f; }
template <typename T> static T* to_dev(const std::vector<T>& h) { T* d; CK(hipMalloc(&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 96, n = argc > 2 ? atoi(argv[2]) : 16, h = argc > 3 ? atoi(argv[3]) : 256, w = argc > 4 ? atoi(argv[4]) : 320;
    const int64_t plane = (int64_t)h * w, pplane = plane / 4;
    printf("transition-down data gradient: N=%d %dx%d C=%d\n", n, h, w, C);
    std::vector<float> hx((size_t)n * C * plane), hdy((size_t)n * C * pplane), hw((size_t)C * C), hg(C), hb(C), hs(2 * C), hold((size_t)n * C * plane);
    std::vector<uint8_t> hidx((size_t)n * C * pplane);
    for (auto& v : hx) v = rnd(-1.f, 1.f);
    for (auto& v : hdy) v = rnd(-1.f, 1.f);
    for (auto& v : hw) v = rnd(-0.1f, 0.1f);
    for (auto& v : hold) v = rnd(-1.f, 1.f);
    for (auto& v : hidx) { rs = rs * 1664525u + 1013904223u; v = (rs >> 16) & 3; }
    for (int c = 0; c < C; ++c) { hg[c] = rnd(0.8f, 1.2f); hb[c] = rnd(-0.1f, 0.1f); hs[2 * c] = 0.01f * (c % 7); hs[2 * c + 1] = 1.7f; }
    float *x = to_dev(hx), *dy = to_dev(hdy), *wgt = to_dev(hw), *gam = to_dev(hg), *bet = to_dev(hb), *saved = to_dev(hs), *oldv = to_dev(hold);
    uint8_t* idx = to_dev(hidx);
    float* out; CK(hipMalloc(&out, hx.size() * sizeof(float)));
    const size_t scr_n = (size_t)2 * C * kBnSlots;
    double* scr; CK(hipMalloc(&scr, scr_n * sizeof(double)));
    ConvParams p{};
    p.n = n; p.h = h; p.w = w;
    p.in = dy; p.in_ns = (int64_t)C * pplane; p.in_cs = (int)pplane; p.in_w = w / 2; p.in_idx = idx; p.idx_ns = (int64_t)C * pplane; p.cin = C;
    p.wgt = wgt; p.w_cout = C; p.w_cin = C;
    p.out = out; p.out_ns = (int64_t)C * plane; p.out_cs = (int)plane; p.out_w = w; p.cout = C;
    p.x = x; p.x_ns = (int64_t)C * plane; p.x_cs = (int)plane;
    p.bn_saved = saved; p.bn_gamma = gam; p.bn_beta = bet; p.bn_scratch = scr; p.bn_slot_stride = 2 * C; p.acc_from = 0;
    int cus = 256; { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount; }
    struct V { std::string name; std::function<int()> run; };
    std::vector<V> vs;
    if ((w / 2) % 4 == 0) vs.push_back({"conv_dma<1,16,2,UNPOOL,DGRAD_BN> (per tile)", [&]() { return launch_conv_dma_auto<1, 16, 2, IN_UNPOOL, EPI_DGRAD_BN, 4>(p, 0); }});
    else vs.push_back({"conv_mfma<1,16,3,UNPOOL,DGRAD_BN> (register-staged)", [&]() { return launch_conv_auto<1, 16, 3, IN_UNPOOL, EPI_DGRAD_BN, 4>(p, 0); }});
    if (td_dgrad_ok(p)) vs.push_back({"td_dgrad persistent, one block per CU", [&]() { return launch_td_dgrad(p, cus, 0); }});
    if (td_dgrad_small_ok(p)) vs.push_back({"td_dgrad_small (128-pixel runs, expanded chunks)", [&]() { return launch_td_dgrad_small(p, 0); }});
    std::vector<float> ref, cur(hx.size());
    std::vector<double> sref, sraw(scr_n);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double flops = 2.0 * n * plane * C * C;
    for (auto& v : vs) {
        CK(hipMemcpy(out, oldv, hx.size() * sizeof(float), hipMemcpyDeviceToDevice));
        CK(hipMemset(scr, 0, scr_n * sizeof(double)));
        int rc = v.run(); if (rc) { printf("%-48s launch failed rc=%d\n", v.name.c_str(), rc); continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(cur.data(), out, cur.size() * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(sraw.data(), scr, scr_n * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<double> s(2 * C, 0.0);
        for (int k = 0; k < kBnSlots; ++k) for (int i = 0; i < 2 * C; ++i) s[i] += sraw[(size_t)k * 2 * C + i];
        if (ref.empty()) { ref = cur; sref = s; }
        double md = 0, mr = 0, sd = 0, sm = 0; size_t worst = 0;
        for (size_t i = 0; i < cur.size(); ++i) { const double d = fabs((double)cur[i] - ref[i]); if (d > md) { md = d; worst = i; } mr = fmax(mr, fabs((double)ref[i])); }
        for (int i = 0; i < 2 * C; ++i) { sd = fmax(sd, fabs(s[i] - sref[i])); sm = fmax(sm, fabs(sref[i])); }
        for (int i = 0; i < 3; ++i) v.run();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rr = 0; rr < 3; ++rr) { CK(hipEventRecord(a, 0)); for (int i = 0; i < 10; ++i) v.run(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = fminf(best, ms / 10); }
        const size_t wi = worst % plane; const int wc = (int)((worst / plane) % C), wn = (int)(worst / plane / C);
        if (md > 1e-3) {          // host evaluation of the worst element
            const int y = (int)(wi / w), xx = (int)(wi % w);
            double da = 0; const size_t pp = (size_t)(y / 2) * (w / 2) + xx / 2; const int wantc = 2 * (y & 1) + (xx & 1);
            for (int o = 0; o < C; ++o) { const size_t q = ((size_t)wn * C + o) * pplane + pp; if (hidx[q] == wantc) da += (double)hw[(size_t)o * C + wc] * hdy[q]; }
            const double xcv = hx[worst] - hs[2 * wc], z = xcv * hg[wc] * hs[2 * wc + 1] + hb[wc], dz = z > 0 ? da : 0;
            printf("   host: old %g dA %g z %g -> out %g\n", hold[worst], da, z, hold[worst] + hg[wc] * hs[2 * wc + 1] * dz);
            size_t bad = 0; for (size_t i = 0; i < cur.size(); ++i) if (fabs((double)cur[i] - ref[i]) > 1e-3) ++bad;
            printf("   %zu of %zu elements differ by more than 1e-3; first ones:", bad, cur.size());
            int shown = 0; for (size_t i = 0; i < cur.size() && shown < 12; ++i) if (fabs((double)cur[i] - ref[i]) > 1e-3) { printf(" (c %d y %zu x %zu)", (int)((i / plane) % C), (i % plane) / w, (i % plane) % w); ++shown; }
            printf("\n");
        }
        printf("%-48s %8.1f us %7.1f TFLOP/s  out diff %.2e / %.2e (worst at n %d c %d y %zu x %zu: %g vs %g)  sums diff %.2e / %.2e\n", v.name.c_str(), best * 1e3, flops / best * 1e-9,
               md, mr, wn, wc, wi / w, wi % w, cur[worst], ref[worst], sd, sm);
    }
    return 0;
}
