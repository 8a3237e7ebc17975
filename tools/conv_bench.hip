// Kernel-variant microbenchmark (development tool, not part of the product): times alternative
// instantiations of the dense-layer kernels on one layer shape and cross-checks their outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/conv_bench.hip -o gpurun_out/conv_bench
//   gpurun_out/conv_bench [cin] [n] [h] [w]
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include <functional>

#include "../endoscopydepthestimation-pytorch_amd/csrc/dgrad_kernels.h"
#include "../endoscopydepthestimation-pytorch_amd/csrc/wgrad_taps_kernels.h"

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
        for (size_t i = 0; i < out_n; ++i) { maxdiff = fmax(maxdiff, fabs((double)cur[i] - ref[i])); maxref = fmax(maxref, fabs((double)ref[i])); }
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

int main(int argc, char** argv) {
    const int cin = argc > 1 ? atoi(argv[1]) : 180;
    const int n = argc > 2 ? atoi(argv[2]) : 8;
    const int h = argc > 3 ? atoi(argv[3]) : 256;
    const int w = argc > 4 ? atoi(argv[4]) : 320;
    const int t = cin + 12;                       // level buffer: cin input planes + 12 output planes
    const int64_t plane = (int64_t)h * w;
    printf("dense layer: N=%d %dx%d Cin=%d -> 12\n", n, h, w, cin);
    float* buf = dev_random((size_t)n * t * plane, -1.f, 1.f, 1);
    float* gbuf = dev_random((size_t)n * t * plane, -1.f, 1.f, 2);
    float* wgt = dev_random((size_t)12 * cin * 9, -0.05f, 0.05f, 3);
    float* bias = dev_random(12, -0.1f, 0.1f, 4);
    float* gamma = dev_random(cin, 0.8f, 1.2f, 5);
    float* beta = dev_random(cin, -0.1f, 0.1f, 6);
    float* rm = dev_random(cin, -0.1f, 0.1f, 7);
    float* rv = dev_random(cin, 0.9f, 1.1f, 8);
    float* saved; CK(hipMalloc(&saved, 2 * cin * sizeof(float)));
    std::vector<float> hs(2 * cin); for (int c = 0; c < cin; ++c) { hs[2 * c] = 0.01f * (c % 7); hs[2 * c + 1] = 1.7f; }
    CK(hipMemcpy(saved, hs.data(), hs.size() * sizeof(float), hipMemcpyHostToDevice));
    double* sums; CK(hipMalloc(&sums, 2 * t * sizeof(double)));
    std::vector<double> hsum(2 * t); const double cnt = (double)n * plane;
    for (int c = 0; c < t; ++c) { hsum[2 * c] = 0.01 * (c % 5) * cnt; hsum[2 * c + 1] = (1.0 / 3.0 + 1e-4 * (c % 5) * (c % 5)) * cnt; }
    CK(hipMemcpy(sums, hsum.data(), hsum.size() * sizeof(double), hipMemcpyHostToDevice));
    double* scratch; CK(hipMalloc(&scratch, 2 * t * sizeof(double))); CK(hipMemset(scratch, 0, 2 * t * sizeof(double)));
    double* osums; CK(hipMalloc(&osums, 2 * 16 * sizeof(double))); CK(hipMemset(osums, 0, 32 * sizeof(double)));
    float* dw; CK(hipMalloc(&dw, (size_t)12 * cin * 9 * sizeof(float)));

    const double flops = 2.0 * n * plane * cin * 12 * 9;

    // ---------------- forward: BN+ReLU -> conv3x3 -> 12 new planes ----------------
    ConvParams f{};
    f.n = n; f.h = h; f.w = w;
    f.in = buf; f.in_ns = t * plane; f.in_cs = (int)plane; f.in_w = w; f.cin = cin;
    f.in_sums = sums; f.gamma = gamma; f.beta = beta; f.running_mean = rm; f.running_var = rv; f.saved = nullptr;
    f.count = cnt; f.eps = 1e-5f; f.momentum = 0.f; f.training = 1;
    f.wgt = wgt; f.bias = bias; f.w_cout = 12; f.w_cin = cin;
    f.out = buf + cin * plane; f.out_ns = t * plane; f.out_cs = (int)plane; f.out_w = w; f.cout = 12; f.out_sums = osums;
    {
        std::vector<Variant> vs;
        vs.push_back({"fwd reg-staged KC8 32x16", [&](hipStream_t s) { return launch_conv<3, 8, 1, IN_BNRELU, EPI_FWD, 2, 8>(f, s); }});
        vs.push_back({"fwd reg-staged KC8 16x16", [&](hipStream_t s) { return launch_conv<3, 8, 1, IN_BNRELU, EPI_FWD, 1, 4>(f, s); }});
        vs.push_back({"fwd reg-staged KC8 32x8", [&](hipStream_t s) { return launch_conv<3, 8, 1, IN_BNRELU, EPI_FWD, 2, 4>(f, s); }});
        vs.push_back({"fwd dma KC4 2buf 32x16 minw5", [&](hipStream_t s) { return launch_conv_dma<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 5>(f, s); }});
        vs.push_back({"fwd dma KC4 2buf 32x16 minw1", [&](hipStream_t s) { return launch_conv_dma<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1>(f, s); }});
        vs.push_back({"fwd dma KC8 2buf 32x16", [&](hipStream_t s) { return launch_conv_dma<3, 8, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1>(f, s); }});
        vs.push_back({"fwd dma KC4 2buf 32x8", [&](hipStream_t s) { return launch_conv_dma<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 4, 2, 1>(f, s); }});
        vs.push_back({"fwd dma KC8 2buf 16x16", [&](hipStream_t s) { return launch_conv_dma<3, 8, 1, IN_BNRELU, EPI_FWD, 1, 4, 2, 1>(f, s); }});
        vs.push_back({"fwd dma4 KC4 32x16 xf0 (bn at frag read)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC4 32x16 xf1 (bn in place)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1, 4, 1>(f, s); }});
        vs.push_back({"fwd dma4 KC4 32x16 xf1 minw5", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 5, 4, 1>(f, s); }});
        vs.push_back({"fwd dma4 KC8 32x16 xf1", [&](hipStream_t s) { return launch_conv_dma_vec<3, 8, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1, 4, 1>(f, s); }});
        vs.push_back({"fwd dma4 KC4 32x8 xf1", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 4, 2, 1, 4, 1>(f, s); }});
        vs.push_back({"fwd dma4 KC4 16x16 xf1", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 1, 4, 2, 1, 4, 1>(f, s); }});
        vs.push_back({"fwd dma1 KC4 32x16 xf1 (dword)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1, 1, 1>(f, s); }});
        vs.push_back({"fwd dma1 KC4 2buf 32x16 (dword)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1, 1, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC4 2buf 32x16 (16B)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC8 2buf 32x16 (16B)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 8, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC4 2buf 32x8 (16B)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 4, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC8 2buf 32x8 (16B)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 8, 1, IN_BNRELU, EPI_FWD, 2, 4, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC4 2buf 16x16 (16B)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 1, 4, 2, 1, 4, 0>(f, s); }});
        bench(vs, f.out, (size_t)12 * plane, flops);      // compares sample 0's 12 planes
    }

    // ---------------- dgrad: dY (12 planes) -> gradient of the cin input planes, BN/ReLU backward fused ----------------
    ConvParams d{};
    d.n = n; d.h = h; d.w = w;
    d.in = gbuf + cin * plane; d.in_ns = t * plane; d.in_cs = (int)plane; d.in_w = w; d.cin = 12;
    d.wgt = wgt; d.w_cout = 12; d.w_cin = cin;
    d.out = gbuf; d.out_ns = t * plane; d.out_cs = (int)plane; d.out_w = w; d.cout = cin;
    d.x = buf; d.x_ns = t * plane; d.x_cs = (int)plane;
    d.bn_saved = saved; d.bn_gamma = gamma; d.bn_beta = beta; d.bn_scratch = scratch; d.acc_from = 1 << 30;   // overwrite: repeatable
    {
        std::vector<Variant> vs;
        vs.push_back({"dgrad reg-staged KC12 Q3 32x8", [&](hipStream_t s) { return launch_conv<3, 12, 3, IN_PLAIN, EPI_DGRAD_BN, 2, 4>(d, s); }});
        vs.push_back({"dgrad reg-staged KC12 Q3 32x16", [&](hipStream_t s) { return launch_conv<3, 12, 3, IN_PLAIN, EPI_DGRAD_BN, 2, 8>(d, s); }});
        vs.push_back({"dgrad dma KC12 Q3 32x8 1buf", [&](hipStream_t s) { return launch_conv_dma<3, 12, 3, IN_PLAIN, EPI_DGRAD_BN, 2, 4, 1, 1>(d, s); }});
        vs.push_back({"dgrad dma KC12 Q3 16x16 1buf", [&](hipStream_t s) { return launch_conv_dma<3, 12, 3, IN_PLAIN, EPI_DGRAD_BN, 1, 4, 1, 1>(d, s); }});
        vs.push_back({"dgrad dma KC12 Q2 32x8 1buf", [&](hipStream_t s) { return launch_conv_dma<3, 12, 2, IN_PLAIN, EPI_DGRAD_BN, 2, 4, 1, 1>(d, s); }});
        vs.push_back({"dgrad dma KC12 Q2 32x16 1buf", [&](hipStream_t s) { return launch_conv_dma<3, 12, 2, IN_PLAIN, EPI_DGRAD_BN, 2, 8, 1, 1>(d, s); }});
        vs.push_back({"dgrad dma KC12 Q1 32x16 1buf", [&](hipStream_t s) { return launch_conv_dma<3, 12, 1, IN_PLAIN, EPI_DGRAD_BN, 2, 8, 1, 1>(d, s); }});
        vs.push_back({"dgrad persistent 32x8", [&](hipStream_t s) { return launch_dgrad_dense<2, 4>(d, s); }});
        vs.push_back({"dgrad persistent 32x16", [&](hipStream_t s) { return launch_dgrad_dense<2, 8>(d, s); }});
        vs.push_back({"dgrad persistent 16x16", [&](hipStream_t s) { return launch_dgrad_dense<1, 4>(d, s); }});
        vs.push_back({"dgrad dma4 Q1 32x8", [&](hipStream_t s) { return launch_conv_dma_vec<3, 12, 1, IN_PLAIN, EPI_DGRAD_BN, 2, 4, 1, 1, 4>(d, s); }});
        vs.push_back({"dgrad dma4 Q1 16x16", [&](hipStream_t s) { return launch_conv_dma_vec<3, 12, 1, IN_PLAIN, EPI_DGRAD_BN, 1, 4, 1, 1, 4>(d, s); }});
        vs.push_back({"dgrad dma4 Q2 16x16", [&](hipStream_t s) { return launch_conv_dma_vec<3, 12, 2, IN_PLAIN, EPI_DGRAD_BN, 1, 4, 1, 1, 4>(d, s); }});
        vs.push_back({"dgrad dma4 Q2 32x8", [&](hipStream_t s) { return launch_conv_dma_vec<3, 12, 2, IN_PLAIN, EPI_DGRAD_BN, 2, 4, 1, 1, 4>(d, s); }});
        vs.push_back({"dgrad dma4 Q3 16x8", [&](hipStream_t s) { return launch_conv_dma_vec<3, 12, 3, IN_PLAIN, EPI_DGRAD_BN, 1, 2, 1, 1, 4>(d, s); }});
        vs.push_back({"dgrad dma1 KC12 Q3 32x8 (dword)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 12, 3, IN_PLAIN, EPI_DGRAD_BN, 2, 4, 1, 1, 1>(d, s); }});
        vs.push_back({"dgrad dma4 KC12 Q3 32x8 (16B)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 12, 3, IN_PLAIN, EPI_DGRAD_BN, 2, 4, 1, 1, 4>(d, s); }});
        vs.push_back({"dgrad dma4 KC12 Q3 32x8 minw2", [&](hipStream_t s) { return launch_conv_dma_vec<3, 12, 3, IN_PLAIN, EPI_DGRAD_BN, 2, 4, 1, 2, 4>(d, s); }});
        vs.push_back({"dgrad dma4 KC12 Q3 32x8 minw3", [&](hipStream_t s) { return launch_conv_dma_vec<3, 12, 3, IN_PLAIN, EPI_DGRAD_BN, 2, 4, 1, 3, 4>(d, s); }});
        bench(vs, d.out, (size_t)cin * plane, flops);
    }

    // ---------------- wgrad ----------------
    WgradParams g{};
    g.n = n; g.h = h; g.w = w; g.tiles_x = (w + kWgTileX - 1) / kWgTileX; g.tiles_y = (h + kWgTileY - 1) / kWgTileY;
    g.in = buf; g.in_ns = t * plane; g.in_cs = (int)plane; g.in_w = w; g.cin = cin;
    g.saved = saved; g.gamma = gamma; g.beta = beta;
    g.dy = gbuf + cin * plane; g.dy_ns = t * plane; g.dy_cs = (int)plane; g.dy_w = w; g.cout = 12;
    g.dw = dw;
    {
        std::vector<Variant> vs;
        vs.push_back({"wgrad KC16 32x8 reg-staged", [&](hipStream_t s) { CK(hipMemsetAsync(dw, 0, (size_t)12 * cin * 9 * 4, s)); return launch_wgrad<3, 1, IN_BNRELU, DY_PLAIN>(g, s); }});
        vs.push_back({"wgrad taps-in-M dma 32x8", [&](hipStream_t s) { CK(hipMemsetAsync(dw, 0, (size_t)12 * cin * 9 * 4, s)); return launch_wgrad_taps<12, IN_BNRELU>(g, s); }});
        bench(vs, dw, (size_t)12 * cin * 9, flops);
    }
    return 0;
}
