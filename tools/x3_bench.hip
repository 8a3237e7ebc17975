// fp32 products on the bf16 matrix cores ("bf16x3", csrc/common.h split_bf16x8): speed and ACCURACY of the dense-layer weight gradient in its
// three operand modes -- fp32 MFMA, operands rounded to bf16 (the mixed-precision mode), three-term split with six bf16 MFMAs -- against an
// fp64 evaluation of the same sums on the host (a sample of the weight-gradient entries) -- and, since round 4, of the Winograd F(3x3, 4x4)
// form (csrc/wgrad_f34_kernels.h) with its diagnostic build without activation loads.  Development tool, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/x3_bench.hip -o tools/bin/x3_bench
//   tools/bin/x3_bench [cin] [n] [h] [w]
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include <functional>

#include "../endoscopydepthestimation-pytorch_amd/csrc/wgrad_taps_kernels.h"
#include "../endoscopydepthestimation-pytorch_amd/csrc/wgrad_nsplit_kernels.h"
#include "../endoscopydepthestimation-pytorch_amd/csrc/wgrad_x3_kernels.h"
#include "../endoscopydepthestimation-pytorch_amd/csrc/wgrad_f34_kernels.h"

using namespace endo;

endo::ProfScope::ProfScope(int f, hipStream_t s, double, double) : family(f), stream(s), slot(nullptr) {}
endo::ProfScope::~ProfScope() {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static std::vector<float> host_random(size_t n, float lo, float hi, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = lo + (hi - lo) * ((s >> 8) & 0xFFFFFF) / 16777215.0f; }
    return h;
}
static float* to_dev(const std::vector<float>& h) {
    float* d; CK(hipMalloc(&d, h.size() * sizeof(float)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    const int cin = argc > 1 ? atoi(argv[1]) : 180;
    const int n = argc > 2 ? atoi(argv[2]) : 16;
    const int h = argc > 3 ? atoi(argv[3]) : 256;
    const int w = argc > 4 ? atoi(argv[4]) : 320;
    const int t = cin + 12;
    const int64_t plane = (int64_t)h * w;
    printf("dense-layer weight gradient: N=%d %dx%d Cin=%d -> 12\n", n, h, w, cin);
    std::vector<float> hx = host_random((size_t)n * t * plane, -1.f, 1.f, 1);
    std::vector<float> hg = host_random((size_t)n * t * plane, -1.f, 1.f, 2);
    std::vector<float> hgamma = host_random(cin, 0.8f, 1.2f, 5), hbeta = host_random(cin, -0.1f, 0.1f, 6);
    std::vector<float> hs(2 * cin); for (int c = 0; c < cin; ++c) { hs[2 * c] = 0.01f * (c % 7); hs[2 * c + 1] = 1.7f; }
    float *buf = to_dev(hx), *gbuf = to_dev(hg), *gamma = to_dev(hgamma), *beta = to_dev(hbeta), *saved = to_dev(hs);
    float* dw; CK(hipMalloc(&dw, (size_t)12 * cin * 9 * sizeof(float)));
    float* wscratch; CK(hipMalloc(&wscratch, kNsScratchFloats * sizeof(float)));
    const double flops = 2.0 * n * plane * cin * 12 * 9;
    WgradParams g{};
    g.n = n; g.h = h; g.w = w; g.tiles_x = (w + kWgTileX - 1) / kWgTileX; g.tiles_y = (h + kWgTileY - 1) / kWgTileY;
    g.in = buf; g.in_ns = t * plane; g.in_cs = (int)plane; g.in_w = w; g.cin = cin;
    g.saved = saved; g.gamma = gamma; g.beta = beta;
    g.dy = gbuf + cin * plane; g.dy_ns = t * plane; g.dy_cs = (int)plane; g.dy_w = w; g.cout = 12;
    g.dw = dw;

    // fp64 reference for a sample of entries: input channels ci = 0, 17, 34, ... (every 17th), all 12 x 9 (co, tap)
    std::vector<int> cis; for (int c = 0; c < cin; c += 17) cis.push_back(c);
    std::vector<double> ref(cis.size() * 108, 0.0);
    for (size_t k = 0; k < cis.size(); ++k) {
        const int ci = cis[k];
        const double mean = hs[2 * ci], scale = (double)(hgamma[ci] * hs[2 * ci + 1]), bt = hbeta[ci];      // scale: the kernel's fp32 product
        std::vector<double> a(plane);
        for (int s = 0; s < n; ++s) {
            const float* xp = hx.data() + ((int64_t)s * t + ci) * plane;
            for (int64_t i = 0; i < plane; ++i) {
                // the kernel's own fp32 BN + ReLU (fmaf(x - mean, scale, beta)), so that the comparison isolates the contraction
                const float v = fmaxf(fmaf(xp[i] - (float)mean, (float)scale, (float)bt), 0.f);
                a[i] = v;
            }
            for (int co = 0; co < 12; ++co) {
                const float* gp = hg.data() + ((int64_t)s * t + cin + co) * plane;
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        double acc = 0.0;
                        for (int y = 0; y < h; ++y) {
                            const int yy = y + ky - 1;
                            if (yy < 0 || yy >= h) continue;
                            const int x0 = kx == 0 ? 1 : 0, x1 = kx == 2 ? w - 1 : w;          // dW[co][ci][ky][kx] = sum_p a[p + (ky-1, kx-1)] dY[co][p]
                            const double* ar = a.data() + (int64_t)yy * w + (kx - 1);
                            const float* gr = gp + (int64_t)y * w;
                            for (int x = x0; x < x1; ++x) acc += ar[x] * (double)gr[x];
                        }
                        ref[k * 108 + co * 9 + ky * 3 + kx] += acc;
                    }
            }
        }
    }
    double maxref = 0; for (double v : ref) maxref = fmax(maxref, fabs(v));

    struct V { const char* name; int mode; };
    const V vs[] = {{"fp32 MFMA (v_mfma_f32_16x16x4_f32)", 0}, {"operands rounded to bf16 (1 x bf16 MFMA)", 1}, {"three-term split per fragment, 6 x bf16 MFMA", 2}, {"three-term split, G pre-split in LDS (wgrad_x3_kernel)", 3}, {"diagnostic: 6 MFMAs on rounded operands, no split", 5}, {"diagnostic: fp32 MFMA, conflict-free (wrong) fragment addresses", 6}, {"Winograd F(3x3, 4x4), fp32 MFMA (wgrad_f34_kernel)", 7}, {"diagnostic: F(3x3, 4x4) without x loads", 9}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> got((size_t)12 * cin * 9);
    for (const V& v : vs) {
        CK(hipMemset(dw, 0, got.size() * 4));
        auto run = [&](int mode) { if (mode == 6) { const int groups = (g.cin + 15) / 16, passes = (groups + 11) / 12, ng = ((groups + passes - 1) / passes + 3) / 4; return ng <= 1 ? launch_wgrad_nsplit_ng<1, 8>(g, wscratch, passes, 0) : ng == 2 ? launch_wgrad_nsplit_ng<2, 8>(g, wscratch, passes, 0) : launch_wgrad_nsplit_ng<3, 8>(g, wscratch, passes, 0); } if (mode == 7) return wgrad_f34_ok(g, 16) ? launch_wgrad_f34(g, wscratch, 0) : -1; if (mode == 9) return launch_wgrad_f34<2>(g, wscratch, 0); return mode == 3 ? launch_wgrad_x3(g, wscratch, 0) : launch_wgrad_nsplit(g, wscratch, 0, mode == 5 ? 3 : mode); };
        int rc = run(v.mode);
        if (rc) { printf("%s: launch failed %d\n", v.name, rc); continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), dw, got.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, sq = 0;
        for (size_t k = 0; k < cis.size(); ++k)
            for (int m = 0; m < 108; ++m) {
                const int co = m / 9, tap = m % 9;
                const double d = got[((size_t)co * cin + cis[k]) * 9 + tap] - ref[k * 108 + m];
                maxerr = fmax(maxerr, fabs(d)); sq += d * d;
            }
        for (int i = 0; i < 3; ++i) run(v.mode);
        CK(hipDeviceSynchronize());
        const int reps = 20;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) run(v.mode);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.1f us  %6.1f TFLOP/s (fp32-equivalent)   vs fp64: max err %.3e, rms %.3e of max |dW| %.3e -> %.2e / %.2e relative\n", v.name, ms / reps * 1e3,
               flops / (ms / reps * 1e-3) / 1e12, maxerr, sqrt(sq / (cis.size() * 108)), maxref, maxerr / maxref, sqrt(sq / (cis.size() * 108)) / maxref);
    }
    return 0;
}
