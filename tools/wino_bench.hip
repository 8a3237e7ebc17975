// Winograd data-gradient microbenchmark (development tool, not part of the product): times dgrad_wino8_kernel and its
// diagnostic / candidate variants on one dense block's base-channel pass and cross-checks their outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize tools/wino_bench.hip -o tools/bin/wino_bench
//   tools/bin/wino_bench [c0] [n] [h] [w]     (c0 = base channels of the block: 48 / 144 at level 0, 96 / 192 at level 1)
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include <functional>

#include "../endoscopydepthestimation-pytorch_amd/csrc/dgrad_wino_kernels.h"
#include "../endoscopydepthestimation-pytorch_amd/csrc/dgrad_wino3_kernels.h"
#include "../endoscopydepthestimation-pytorch_amd/csrc/dgrad_wino3p_kernels.h"

using namespace endo;

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

static void bench(std::vector<Variant>& vs, float* out, size_t out_n, double* scratch, size_t scratch_n, double flops) {
    std::vector<float> ref, cur(out_n);
    std::vector<double> sref, scur, sraw(scratch_n);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto& v : vs) {
        CK(hipMemset(out, 0, out_n * sizeof(float)));
        CK(hipMemset(scratch, 0, scratch_n * sizeof(double)));
        int rc = v.run(0);
        if (rc) { printf("%-52s launch failed rc=%d\n", v.name.c_str(), rc); continue; }
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("%-52s FAILED: %s\n", v.name.c_str(), hipGetErrorString(e)); exit(1); }
        CK(hipMemcpy(cur.data(), out, out_n * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(sraw.data(), scratch, scratch_n * sizeof(double), hipMemcpyDeviceToHost));
        double maxdiff = 0, maxref = 0, sdiff = 0, smax = 0;
        {   // the BN sums: [layer][slot copy][value] -> add the copies up
            const size_t per_layer = scratch_n / 4, per_copy = per_layer / kBnSlots;
            std::vector<double> folded(4 * per_copy, 0.0);
            for (int l = 0; l < 4; ++l) for (int c = 0; c < kBnSlots; ++c) for (size_t i = 0; i < per_copy; ++i) folded[l * per_copy + i] += sraw[l * per_layer + c * per_copy + i];
            scur = folded;
        }
        if (ref.empty()) { ref = cur; sref = scur; }
        for (size_t i = 0; i < out_n; ++i) { maxdiff = fmax(maxdiff, fabs((double)cur[i] - ref[i])); maxref = fmax(maxref, fabs((double)ref[i])); }
        for (size_t i = 0; i < sref.size(); ++i) { sdiff = fmax(sdiff, fabs(scur[i] - sref[i])); smax = fmax(smax, fabs(sref[i])); }
        for (int i = 0; i < 3; ++i) v.run(0);
        CK(hipDeviceSynchronize());
        const int reps = 10;
        float best = 1e30f, tot = 0.f;
        for (int rr = 0; rr < 3; ++rr) {
            CK(hipEventRecord(a, 0));
            for (int i = 0; i < reps; ++i) v.run(0);
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            best = fminf(best, ms / reps); tot += ms / reps;
        }
        printf("%-52s %8.1f us (best %8.1f)  %6.1f TFLOP/s   out diff %.2e / %.2e   sums diff %.2e / %.2e\n", v.name.c_str(), tot / 3 * 1e3, best * 1e3,
               flops / (tot / 3 * 1e-3) / 1e12, maxdiff, maxref, sdiff, smax);
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    const int c0 = argc > 1 ? atoi(argv[1]) : 144;
    const int n = argc > 2 ? atoi(argv[2]) : 16;
    const int h = argc > 3 ? atoi(argv[3]) : 256;
    const int w = argc > 4 ? atoi(argv[4]) : 320;
    int mode = argc > 5 ? atoi(argv[5]) : 0;          // 0 = everything, 1 = product candidates only, 2 = direct / wino3 / persistent forms compared over ALL samples
    const int64_t plane = (int64_t)h * w;
    const int cin_last = c0 + 36;
    printf("dense block base pass: N=%d %dx%d C0=%d, 4 layers x 12 dY maps\n", n, h, w, c0);
    float* xbuf = dev_random((size_t)n * c0 * plane, -1.f, 1.f, 1);
    float* gbuf = dev_random((size_t)n * 48 * plane, -1.f, 1.f, 2);
    float* obuf; CK(hipMalloc(&obuf, (size_t)n * c0 * plane * sizeof(float)));
    float* gamma = dev_random(cin_last, 0.8f, 1.2f, 5);
    float* beta = dev_random(cin_last, -0.1f, 0.1f, 6);
    float* saved; CK(hipMalloc(&saved, 2 * cin_last * sizeof(float)));
    std::vector<float> hs(2 * cin_last); for (int c = 0; c < cin_last; ++c) { hs[2 * c] = 0.01f * (c % 7); hs[2 * c + 1] = 1.7f; }
    CK(hipMemcpy(saved, hs.data(), hs.size() * sizeof(float), hipMemcpyHostToDevice));
    const size_t scratch_n = (size_t)4 * 2 * cin_last * kBnSlots;
    double* scratch; CK(hipMalloc(&scratch, scratch_n * sizeof(double)));
    // four layers' weights W[12][cin_l][3][3], cin_l = c0 + 12 l
    size_t woff[4], wtot = 0;
    for (int l = 0; l < 4; ++l) { woff[l] = wtot; wtot += (size_t)12 * (c0 + 12 * l) * 9; }
    float* wgt = dev_random(wtot, -0.05f, 0.05f, 3);

    WinoDgradTable tb{};
    tb.layers = 4;
    int64_t uoff = 0; int start = 0;
    for (int l = 0; l < 4; ++l) {
        tb.start[l] = start; tb.cin[l] = c0 + 12 * l; tb.groups[l] = c0 / 16; tb.w_off[l] = woff[l]; tb.u_off[l] = uoff;
        start += 16 * tb.groups[l] * 12; uoff += (int64_t)tb.groups[l] * kWinoDgradSlice;
    }
    tb.start[4] = start;
    float* ubuf; CK(hipMalloc(&ubuf, uoff * sizeof(float)));
    float* ubuf1; CK(hipMalloc(&ubuf1, uoff * sizeof(float)));
    dgrad_wino_weights_kernel<<<(start + 255) / 256, 256>>>(tb, wgt, ubuf, 0);
    dgrad_wino_weights_kernel<<<(start + 255) / 256, 256>>>(tb, wgt, ubuf1, 1 << 20);
    CK(hipDeviceSynchronize());

    DgradBlockParams p{};
    p.n = n; p.h = h; p.w = w;
    p.g = gbuf; p.g_ns = 48 * plane; p.g_cs = (int)plane; p.g_w = w;
    p.x = xbuf; p.out = obuf; p.ns = c0 * plane; p.cs = (int)plane; p.count = c0; p.acc_from = 1 << 30; p.w_ci_off = 0;
    for (int j = 0; j < 4; ++j) {
        p.wgt[j] = wgt + woff[j]; p.w_cin[j] = c0 + 12 * j; p.saved[j] = saved; p.gamma[j] = gamma; p.beta[j] = beta;
        p.scratch[j] = scratch + (size_t)j * 2 * cin_last * kBnSlots;
    }
    p.slot_stride = 2 * cin_last;
    p.group_n = 0; p.gs = 0;
    const float* const u[4] = {ubuf + tb.u_off[0], ubuf + tb.u_off[1], ubuf + tb.u_off[2], ubuf + tb.u_off[3]};
    const float* const u1[4] = {ubuf1 + tb.u_off[0], ubuf1 + tb.u_off[1], ubuf1 + tb.u_off[2], ubuf1 + tb.u_off[3]};
    const double flops = 2.0 * n * plane * c0 * 12 * 9 * 4;

    std::vector<Variant> vs;
    vs.push_back({"dgrad_block8<4> direct (reference values)", [&](hipStream_t s) { return launch_dgrad_block8<4>(p, s); }});
    vs.push_back({"dgrad_wino8<4> (round-2 library)", [&](hipStream_t s) { return launch_dgrad_wino8<4>(p, u, s); }});
    vs.push_back({"dgrad_wino3<4> (library)", [&](hipStream_t s) { return launch_dgrad_wino3<4, 0, 0>(p, u1, s); }});
    // persistent blocks (dgrad_wino3p_kernels.h): one block per CU walking a run of tiles
    int cus = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount; }
    vs.push_back({"dgrad_wino3p<4> persistent, one block per CU", [&](hipStream_t s) { return launch_dgrad_wino3p<4, false, 0>(p, u1, cus, nullptr, nullptr, s); }});
    vs.push_back({"wino3p no atomics (4)", [&](hipStream_t s) { return launch_dgrad_wino3p<4, false, 4>(p, u1, cus, nullptr, nullptr, s); }});
    vs.push_back({"wino3p no dY tile loads (8)", [&](hipStream_t s) { return launch_dgrad_wino3p<4, false, 8>(p, u1, cus, nullptr, nullptr, s); }});
    vs.push_back({"wino3p no mem (15)", [&](hipStream_t s) { return launch_dgrad_wino3p<4, false, 15>(p, u1, cus, nullptr, nullptr, s); }});
    if (mode == 5 || mode == 6) {
        // phase lengths of the persistent kernel: block 0 stamps the clock behind every barrier of its third tile (EXP 16); mode 6: old gradient from the buffer
        if (mode == 6) p.acc_from = 0;
        unsigned long long* dbg; CK(hipMalloc(&dbg, 128 * 8)); CK(hipMemset(dbg, 0, 128 * 8));
        auto dump = [&](const char* name) {
            CK(hipDeviceSynchronize());
            unsigned long long h[128]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            printf("%s\n", name);
            for (int w = 0; w < 2; ++w) {
                printf("  worker %d, cycles between barriers:", w);
                for (int i = 1; i < 64 && h[w * 64 + i]; ++i) printf(" %llu", h[w * 64 + i] - h[w * 64 + i - 1]);
                int last = 0; for (int i = 0; i < 64 && h[w * 64 + i]; ++i) last = i;
                printf("  | total %llu over %d intervals\n", h[w * 64 + last] - h[w * 64], last);
            }
        };
        CK(hipMemset(p.out, 0, (size_t)n * c0 * plane * sizeof(float)));
        launch_dgrad_wino3p<4, false, 16>(p, u1, cus, reinterpret_cast<double*>(dbg), nullptr, 0); dump("persistent, full");
        launch_dgrad_wino3p<4, false, 16 + 8>(p, u1, cus, reinterpret_cast<double*>(dbg), nullptr, 0); dump("persistent, no dY tile loads");
        launch_dgrad_wino3p<4, false, 16 + 4>(p, u1, cus, reinterpret_cast<double*>(dbg), nullptr, 0); dump("persistent, no sums");
        launch_dgrad_wino3p<4, false, 16 + 1>(p, u1, cus, reinterpret_cast<double*>(dbg), nullptr, 0); dump("persistent, no x / gradient loads");
        launch_dgrad_wino3p<4, false, 16 + 2>(p, u1, cus, reinterpret_cast<double*>(dbg), nullptr, 0); dump("persistent, no stores");
        launch_dgrad_wino3p<4, false, 16 + 15>(p, u1, cus, reinterpret_cast<double*>(dbg), nullptr, 0); dump("persistent, no mem");
        return 0;
    }
    if (mode == 4) { p.acc_from = 0; mode = 2; }          // the gradient buffer's content is the old gradient (zeroed before the compared run)
    if (mode == 2) { vs.erase(vs.begin() + 1, vs.begin() + 2); bench(vs, p.out, (size_t)n * c0 * plane, scratch, scratch_n, flops); return 0; }
    if (mode == 3) {
        // the virtual final gradient (old gradient = g * vw[channel]) and, in the persistent kernel, the final convolution's weight gradient sum g * x
        float* vg = dev_random((size_t)n * plane, -1.f, 1.f, 11);
        float* vw = dev_random(c0, -1.f, 1.f, 12);
        DgradBlockParams pv = p; pv.vg = vg; pv.vw = vw; pv.acc_from = 0;
        double* fwp; CK(hipMalloc(&fwp, (size_t)cus * c0 * sizeof(double))); CK(hipMemset(fwp, 0, (size_t)cus * c0 * sizeof(double)));
        int used = 0;
        std::vector<Variant> v3;
        v3.push_back({"dgrad_wino3<4> (library), virtual old gradient", [&](hipStream_t s) { return launch_dgrad_wino3<4, 0, 0>(pv, u1, s); }});
        v3.push_back({"dgrad_wino3p<4, FW> persistent, virtual old gradient", [&](hipStream_t s) { return launch_dgrad_wino3p<4, true, 0>(pv, u1, cus, fwp, &used, s); }});
        bench(v3, p.out, (size_t)n * c0 * plane, scratch, scratch_n, flops);
        std::vector<double> parts((size_t)used * c0);
        CK(hipMemcpy(parts.data(), fwp, parts.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::vector<float> hx((size_t)n * c0 * plane), hg((size_t)n * plane);
        CK(hipMemcpy(hx.data(), xbuf, hx.size() * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hg.data(), vg, hg.size() * sizeof(float), hipMemcpyDeviceToHost));
        double worst = 0, big = 0;
        for (int c = 0; c < c0; ++c) {
            double ref = 0, got = 0;
            for (int s = 0; s < n; ++s) for (size_t i = 0; i < plane; ++i) ref += (double)hg[s * plane + i] * hx[((size_t)s * c0 + c) * plane + i];
            for (int b = 0; b < used; ++b) got += parts[(size_t)b * c0 + c];
            worst = fmax(worst, fabs(got - ref)); big = fmax(big, fabs(ref));
        }
        printf("final-conv weight gradient from %d block partials: max |diff| %.3e of max |ref| %.3e\n", used, worst, big);
        return 0;
    }
    vs.push_back({"dgrad_wino3<4> OPT16 (weight DMA at V end)", [&](hipStream_t s) { return launch_dgrad_wino3<4, 0, 16>(p, u1, s); }});
    vs.push_back({"dgrad_wino3<4> OPT32 (setprio 1 in M phases)", [&](hipStream_t s) { return launch_dgrad_wino3<4, 0, 32>(p, u1, s); }});
    vs.push_back({"wino3 no dY tile load (8)", [&](hipStream_t s) { return launch_dgrad_wino3<4, 8, 0>(p, u1, s); }});
    vs.push_back({"wino3 no atomics (4)", [&](hipStream_t s) { return launch_dgrad_wino3<4, 4, 0>(p, u1, s); }});
    vs.push_back({"wino3 no tile load, no atomics (12)", [&](hipStream_t s) { return launch_dgrad_wino3<4, 12, 0>(p, u1, s); }});
    vs.push_back({"wino3 no mem (7)", [&](hipStream_t s) { return launch_dgrad_wino3<4, 7, 0>(p, u1, s); }});
    vs.push_back({"wino3 no mem, no V arithmetic (71)", [&](hipStream_t s) { return launch_dgrad_wino3<4, 71, 0>(p, u1, s); }});
    vs.push_back({"wino3 no mem, no M phase (135)", [&](hipStream_t s) { return launch_dgrad_wino3<4, 135, 0>(p, u1, s); }});
    vs.push_back({"wino3 no mem, neither (199)", [&](hipStream_t s) { return launch_dgrad_wino3<4, 199, 0>(p, u1, s); }});
    if (mode == 0) {
        vs.push_back({"wino8 no atomics (4)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 4>(p, u, s); }});
        vs.push_back({"wino8 no stores (2)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 2>(p, u, s); }});
        vs.push_back({"wino8 no x loads (1)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 1>(p, u, s); }});
        vs.push_back({"wino8 no atomics/stores/loads (7)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 7>(p, u, s); }});
        vs.push_back({"wino8 7 + weights once (15)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 15>(p, u, s); }});
        vs.push_back({"wino8 15 + no barrier (31)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 31>(p, u, s); }});
        vs.push_back({"wino8 31 + trivial epilogue (159)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 159>(p, u, s); }});
        vs.push_back({"wino8 31 + no transform (95)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 95>(p, u, s); }});
        vs.push_back({"wino8 31 + no MFMA (63)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 63>(p, u, s); }});
        vs.push_back({"wino8 MFMA only-ish (223)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 223>(p, u, s); }});
        vs.push_back({"wino8 trivial epilogue only (128)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 128>(p, u, s); }});
        vs.push_back({"wino8 no transform only (64)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 64>(p, u, s); }});
        vs.push_back({"wino8 no barrier only (16)", [&](hipStream_t s) { return launch_dgrad_wino8<4, 16>(p, u, s); }});
    }
    bench(vs, p.out, (size_t)c0 * plane, scratch, scratch_n, flops);
    return 0;
}
