// Transition-down data-gradient microbenchmark (development tool): conv_dma_kernel<1, 16, 2, IN_UNPOOL, EPI_DGRAD_BN> (one block per tile and
// 32 output channels) against td_dgrad_kernel (persistent blocks, td_dgrad_kernels.h); cross-checks outputs and BN-backward sums.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/td_bench.hip -o tools/bin/td_bench
//   tools/bin/td_bench [C] [n] [h] [w] [f]  (C = 96 at level 0, 144 at level 1; h, w = the full-resolution side; f = 1: the FORWARD kernels --
//   conv_dma_kernel<1, 8, 3, IN_BNRELU, EPI_FWD_POOL> against td_fwd_kernel)
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include <functional>
#include "../endoscopydepthestimation-pytorch_amd/csrc/td_dgrad_kernels.h"
#include "../endoscopydepthestimation-pytorch_amd/csrc/td_fwd_kernels.h"
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
    if (argc > 5 && atoi(argv[5]) == 1) {
        // ---- forward: BN (batch statistics from the sums) -> ReLU -> conv1x1 -> maxpool2 with argmax codes and output statistics ----
        std::vector<double> hsum(2 * C);
        for (int c = 0; c < C; ++c) { hsum[2 * c] = 0.02 * (c % 5) * n * plane; hsum[2 * c + 1] = (0.35 + 0.0004 * (c % 5) * (c % 5)) * n * plane; }
        std::vector<float> hrun(2 * C, 0.5f), hbias(C);
        for (auto& v : hbias) v = rnd(-0.1f, 0.1f);
        double* insums = to_dev(hsum); float* run = to_dev(hrun); float* bias = to_dev(hbias);
        float* savedf; CK(hipMalloc(&savedf, 2 * C * sizeof(float)));
        float* pout; CK(hipMalloc(&pout, (size_t)n * C * pplane * sizeof(float)));
        uint8_t* pidx; CK(hipMalloc(&pidx, (size_t)n * C * pplane));
        double* osums; CK(hipMalloc(&osums, 2 * C * sizeof(double)));
        ConvParams q{};
        q.n = n; q.h = h; q.w = w;
        q.in = x; q.in_ns = (int64_t)C * plane; q.in_cs = (int)plane; q.in_w = w; q.cin = C;
        q.in_sums = insums; q.gamma = gam; q.beta = bet; q.running_mean = run; q.running_var = run + C; q.saved = savedf; q.count = (double)n * plane;
        q.eps = 1e-5f; q.momentum = 0.1f; q.training = 1;
        q.wgt = wgt; q.bias = bias; q.w_cout = C; q.w_cin = C;
        q.out = pout; q.out_ns = (int64_t)C * pplane; q.out_cs = (int)pplane; q.out_w = w / 2; q.cout = C;
        q.out_idx = pidx; q.idx_ns = (int64_t)C * pplane; q.out_sums = osums;
        int cus2 = 256; { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus2 = prop.multiProcessorCount; }
        struct V { std::string name; std::function<int()> run; };
        std::vector<V> vs;
        vs.push_back({"conv_dma<1,8,3,BNRELU,FWD_POOL> (per tile)", [&]() { return launch_conv_dma_auto<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 4>(q, 0); }});
        if (td_fwd_ok(q)) vs.push_back({"td_fwd persistent, one block per CU", [&]() { return launch_td_fwd(q, cus2, 0); }});
        if (td_fwd_ok(q) && C == 96) {
            vs.push_back({"td_fwd 8 waves, one block per CU", [&]() { return launch_td_fwd_t<96, 8>(q, cus2, 0); }});
            vs.push_back({"td_fwd 4 waves, two blocks per CU", [&]() { return launch_td_fwd_t<96, 4>(q, cus2, 0); }});
            vs.push_back({"td_fwd<4> no input DMA (1)", [&]() { return launch_td_fwd_t<96, 4, 1>(q, cus2, 0); }});
            vs.push_back({"td_fwd<4> no epilogue (2)", [&]() { return launch_td_fwd_t<96, 4, 2>(q, cus2, 0); }});
            vs.push_back({"td_fwd<4> no MFMAs (4)", [&]() { return launch_td_fwd_t<96, 4, 4>(q, cus2, 0); }});
            vs.push_back({"td_fwd<4> MFMAs + LDS reads only (11)", [&]() { return launch_td_fwd_t<96, 4, 11>(q, cus2, 0); }});
        }
        std::vector<float> ref, cur((size_t)n * C * pplane);
        std::vector<uint8_t> iref, icur((size_t)n * C * pplane);
        std::vector<double> sref, scur(2 * C);
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        const double flops = 2.0 * n * plane * C * C;
        for (auto& v : vs) {
            CK(hipMemset(pout, 0, cur.size() * sizeof(float))); CK(hipMemset(pidx, 0, icur.size())); CK(hipMemset(osums, 0, 2 * C * sizeof(double)));
            CK(hipMemcpy(run, hrun.data(), 2 * C * sizeof(float), hipMemcpyHostToDevice));
            int rc = v.run(); if (rc) { printf("%-48s launch failed rc=%d\n", v.name.c_str(), rc); continue; }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(cur.data(), pout, cur.size() * sizeof(float), hipMemcpyDeviceToHost));
            CK(hipMemcpy(icur.data(), pidx, icur.size(), hipMemcpyDeviceToHost));
            CK(hipMemcpy(scur.data(), osums, 2 * C * sizeof(double), hipMemcpyDeviceToHost));
            if (ref.empty()) { ref = cur; iref = icur; sref = scur; }
            double md = 0, mr = 0, sd = 0, sm = 0; size_t codes = 0;
            for (size_t i = 0; i < cur.size(); ++i) { md = fmax(md, fabs((double)cur[i] - ref[i])); mr = fmax(mr, fabs((double)ref[i])); codes += icur[i] != iref[i]; }
            for (int i = 0; i < 2 * C; ++i) { sd = fmax(sd, fabs(scur[i] - sref[i])); sm = fmax(sm, fabs(sref[i])); }
            for (int i = 0; i < 3; ++i) v.run();
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rr = 0; rr < 3; ++rr) { CK(hipEventRecord(a, 0)); for (int i = 0; i < 10; ++i) v.run(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); best = fminf(best, ms / 10); }
            printf("%-48s %8.1f us %7.1f TFLOP/s  pooled out diff %.2e / %.2e  argmax codes that differ %zu of %zu  sums diff %.2e / %.2e\n", v.name.c_str(), best * 1e3,
                   flops / best * 1e-9, md, mr, codes, cur.size(), sd, sm);
        }
        return 0;
    }
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
