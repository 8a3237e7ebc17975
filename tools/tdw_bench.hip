// The transition-down 1x1 weight gradient stand-alone (csrc/wgrad1x1_kernels.h): the 16-byte-DMA kernel against round 5's dword-DMA form
// (tools/scratch/wgrad1x1_old_kernels.h, a renamed copy kept for this comparison) -- time and the difference of the two dW.  Development tool.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/tdw_bench.hip -o tools/bin/tdw_bench
//   tools/bin/tdw_bench [c] [n] [h] [w]
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#include "../endoscopydepthestimation-pytorch_amd/csrc/wgrad1x1_kernels.h"
#include "scratch/wgrad1x1_old_kernels.h"

using namespace endo;

endo::ProfScope::ProfScope(int f, hipStream_t s, double, double) : family(f), stream(s), slot(nullptr) {}
endo::ProfScope::~ProfScope() {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned rng_state = 12345u;
static float rnd(float lo, float hi) { rng_state = rng_state * 1664525u + 1013904223u; return lo + (hi - lo) * ((rng_state >> 8) & 0xFFFFFF) / 16777215.0f; }

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 96, N = argc > 2 ? atoi(argv[2]) : 16, H = argc > 3 ? atoi(argv[3]) : 256, W = argc > 4 ? atoi(argv[4]) : 320;
    const int h2 = H / 2, w2 = W / 2;
    const size_t plane = static_cast<size_t>(H) * W, plane2 = static_cast<size_t>(h2) * w2;
    std::vector<float> hx(N * C * plane), hg(N * C * plane2), hsaved(2 * C), hgamma(C), hbeta(C);
    std::vector<uint8_t> hidx(N * C * plane2);
    for (auto& v : hx) v = rnd(-1.f, 1.f);
    for (auto& v : hg) v = rnd(-1.f, 1.f);
    for (auto& v : hidx) { rng_state = rng_state * 1664525u + 1013904223u; v = (rng_state >> 13) & 3; }
    for (int c = 0; c < C; ++c) { hsaved[2 * c] = rnd(-0.2f, 0.2f); hsaved[2 * c + 1] = rnd(0.8f, 1.5f); hgamma[c] = rnd(0.5f, 1.5f); hbeta[c] = rnd(-0.3f, 0.3f); }
    float *dx, *dg, *dsaved, *dgamma, *dbeta, *dw0, *dw1; uint8_t* didx;
    CK(hipMalloc(&dx, hx.size() * 4 + 4096)); CK(hipMalloc(&dg, hg.size() * 4 + 4096)); CK(hipMalloc(&didx, hidx.size() + 4096));
    CK(hipMalloc(&dsaved, 2 * C * 4)); CK(hipMalloc(&dgamma, C * 4)); CK(hipMalloc(&dbeta, C * 4)); CK(hipMalloc(&dw0, C * C * 4)); CK(hipMalloc(&dw1, C * C * 4));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(didx, hidx.data(), hidx.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dsaved, hsaved.data(), 2 * C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dgamma, hgamma.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbeta, hbeta.data(), C * 4, hipMemcpyHostToDevice));

    WgradParams p{};
    p.n = N; p.h = H; p.w = W;
    p.in = dx; p.in_ns = static_cast<int64_t>(C) * plane; p.in_cs = static_cast<int>(plane); p.in_w = W; p.cin = C;
    p.saved = dsaved; p.gamma = dgamma; p.beta = dbeta;
    p.dy = dg; p.dy_ns = static_cast<int64_t>(C) * plane2; p.dy_cs = static_cast<int>(plane2); p.dy_w = w2; p.cout = C;
    p.dy_idx = didx; p.idx_ns = static_cast<int64_t>(C) * plane2;
    p.group_n = 0; p.gs = 0; p.in_gs = 0;

    printf("transition-down weight gradient: N=%d %dx%d C=%d   new ok %d, old ok %d\n", N, H, W, C, (int)wgrad1x1_dma_ok(p), (int)wgrad1x1_dma_old_ok(p));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](auto&& launch, float* dw, const char* name) {
        p.dw = dw;
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0.f;
        const int reps = 20;
        for (int i = 0; i < reps; ++i) {
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms); sum += ms;
        }
        CK(hipMemset(dw, 0, C * C * 4)); launch(); CK(hipDeviceSynchronize());
        const double flop = 2.0 * C * C * N * plane;
        printf("%-44s %8.1f us (best %8.1f)  %6.1f TFLOP/s\n", name, 1e3 * sum / reps, 1e3 * best, flop / (sum / reps * 1e-3) * 1e-12);
    };
    time_it([&] { launch_wgrad1x1_dma_old<0>(p, 0); }, dw0, "wgrad1x1_dma (round 5: dword DMAs)");
    time_it([&] { launch_wgrad1x1_dma<0>(p, 0, 2); }, dw1, "16-byte DMAs, 3 x 1 waves (144 x 48 tiles)");
    time_it([&] { launch_wgrad1x1_dma<0>(p, 0, 1); }, dw1, "16-byte DMAs, 2 x 2 waves (96 x 96 tiles)");
    {   // what the final atomics cost: the same launch with p.dw = nullptr (the sums stay in registers: the compiler keeps the MFMAs, the atomics are predicated off)
        hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
        WgradParams q = p; q.dw = nullptr;
        for (int i = 0; i < 3; ++i) launch_wgrad1x1_dma<0>(q, 0);
        float sum = 0.f;
        for (int i = 0; i < 20; ++i) { CK(hipEventRecord(a0, 0)); launch_wgrad1x1_dma<0>(q, 0); CK(hipEventRecord(a1, 0)); CK(hipEventSynchronize(a1)); float ms; CK(hipEventElapsedTime(&ms, a0, a1)); sum += ms; }
        printf("%-44s %8.1f us\n", "library's choice without the final atomics", 1e3 * sum / 20);
    }
    time_it([&] { launch_wgrad1x1_dma<0>(p, 0); }, dw1, "wgrad1x1_dma (library's choice)");
    std::vector<float> h0(C * C), h1(C * C);
    CK(hipMemcpy(h0.data(), dw0, C * C * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), dw1, C * C * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (int i = 0; i < C * C; ++i) { md = fmax(md, fabs(static_cast<double>(h0[i]) - h1[i])); mx = fmax(mx, fabs(h0[i])); }
    printf("max |dW new - dW old| %.3e of max |dW| %.3e\n", md, mx);
    // a sample of entries in fp64 on the host
    double worst = 0;
    for (int t = 0; t < 6; ++t) {
        const int co = (t * 37 + 5) % C, ci = (t * 53 + 11) % C;
        const double sc = static_cast<double>(hgamma[ci] * hsaved[2 * ci + 1]);
        const double sh = static_cast<double>(fmaf(-hsaved[2 * ci], hgamma[ci] * hsaved[2 * ci + 1], hbeta[ci]));
        double s = 0;
        for (int n = 0; n < N; ++n)
            for (int y = 0; y < h2; ++y)
                for (int x = 0; x < w2; ++x) {
                    const size_t q = (static_cast<size_t>(n) * C + co) * plane2 + static_cast<size_t>(y) * w2 + x;
                    const int code = hidx[q];
                    const size_t px = (static_cast<size_t>(n) * C + ci) * plane + static_cast<size_t>(2 * y + (code >> 1)) * W + 2 * x + (code & 1);
                    const double a = fmax(sc * hx[px] + sh, 0.0);
                    s += static_cast<double>(hg[q]) * a;
                }
        worst = fmax(worst, fabs(s - h1[co * C + ci]));
    }
    printf("new kernel against fp64 on 6 entries: max |diff| %.3e\n", worst);
    return 0;
}
