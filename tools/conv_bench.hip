// Kernel-variant microbenchmark (development tool, not part of the product): times alternative
// instantiations of the dense-layer kernels on one layer shape and cross-checks their outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics tools/conv_bench.hip -o tools/bin/conv_bench
//   tools/bin/conv_bench [cin] [n] [h] [w]     (cin = channels entering the LAST layer of a dense block: c0 = cin - 36)
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
#include "../endoscopydepthestimation-pytorch_amd/csrc/wino_fwd_kernels.h"

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
    float* wscratch; CK(hipMalloc(&wscratch, kNsScratchFloats * sizeof(float)));

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
        vs.push_back({"fwd dma4 KC4 2buf 32x16 (library)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd 32x16 only first chunk DMA'd (EXP 1)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1, 4, 0, 1>(f, s); }});
        vs.push_back({"fwd 32x16 no BN transform (EXP 2)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1, 4, 0, 2>(f, s); }});
        vs.push_back({"fwd 32x16 DMA issued but not waited (EXP 4)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1, 4, 0, 4>(f, s); }});
        vs.push_back({"fwd 32x16 neither (EXP 3)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 8, 2, 1, 4, 0, 3>(f, s); }});
        vs.push_back({"fwd dma4 KC4 2buf 16x16", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 1, 4, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC8 2buf 16x16", [&](hipStream_t s) { return launch_conv_dma_vec<3, 8, 1, IN_BNRELU, EPI_FWD, 1, 4, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC4 2buf 32x8", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 4, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC8 2buf 32x8", [&](hipStream_t s) { return launch_conv_dma_vec<3, 8, 1, IN_BNRELU, EPI_FWD, 2, 4, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC4 2buf 16x8", [&](hipStream_t s) { return launch_conv_dma_vec<3, 4, 1, IN_BNRELU, EPI_FWD, 1, 2, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC8 2buf 16x8", [&](hipStream_t s) { return launch_conv_dma_vec<3, 8, 1, IN_BNRELU, EPI_FWD, 1, 2, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC16 2buf 16x8", [&](hipStream_t s) { return launch_conv_dma_vec<3, 16, 1, IN_BNRELU, EPI_FWD, 1, 2, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd KC16 16x8 only first chunk DMA'd (EXP 1)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 16, 1, IN_BNRELU, EPI_FWD, 1, 2, 2, 1, 4, 0, 1>(f, s); }});
        vs.push_back({"fwd KC16 16x8 no BN transform (EXP 2)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 16, 1, IN_BNRELU, EPI_FWD, 1, 2, 2, 1, 4, 0, 2>(f, s); }});
        vs.push_back({"fwd KC16 16x8 DMA not waited (EXP 4)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 16, 1, IN_BNRELU, EPI_FWD, 1, 2, 2, 1, 4, 0, 4>(f, s); }});
        vs.push_back({"fwd KC16 16x8 neither (EXP 3)", [&](hipStream_t s) { return launch_conv_dma_vec<3, 16, 1, IN_BNRELU, EPI_FWD, 1, 2, 2, 1, 4, 0, 3>(f, s); }});
        vs.push_back({"fwd dma4 KC16 2buf 16x16", [&](hipStream_t s) { return launch_conv_dma_vec<3, 16, 1, IN_BNRELU, EPI_FWD, 1, 4, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC8 2buf 16x4", [&](hipStream_t s) { return launch_conv_dma_vec<3, 8, 1, IN_BNRELU, EPI_FWD, 1, 1, 2, 1, 4, 0>(f, s); }});
        vs.push_back({"fwd dma4 KC16 2buf 16x4", [&](hipStream_t s) { return launch_conv_dma_vec<3, 16, 1, IN_BNRELU, EPI_FWD, 1, 1, 2, 1, 4, 0>(f, s); }});
        // what the BN statistics of the stored values cost (sum / sum^2 per channel: shuffles, LDS, and one fp64 atomic per value and block
        // on the two cache lines of the layer's 12 pairs -- blocks of a single-round launch all end together)
        ConvParams fns = f; fns.out_sums = nullptr;
        vs.push_back({"fwd dma4 KC16 2buf 16x8, no output statistics", [&](hipStream_t s) { return launch_conv_dma_vec<3, 16, 1, IN_BNRELU, EPI_FWD, 1, 2, 2, 1, 4, 0>(fns, s); }});
        {
            WinoWeightTable wt{}; wt.layers = 1; wt.start[0] = 0; wt.start[1] = cin * 16; wt.cin[0] = cin; wt.cout[0] = 12; wt.w_off[0] = 0; wt.u_off[0] = 0;
            float* ubuf; CK(hipMalloc(&ubuf, (size_t)cin * kWinoUStride * sizeof(float)));
            wino_fwd_weights_kernel<<<(cin * 16 + 255) / 256, 256>>>(wt, wgt, ubuf);
            CK(hipDeviceSynchronize());
            static ConvParams fw, fwn;
            fw = f; fw.wgt = ubuf; fwn = fw; fwn.out_sums = nullptr;
            if (wino_fwd_ok(fw)) {
                vs.push_back({"fwd Winograd F(2x2,3x3) 32x8 (level-1 form)", [&](hipStream_t s) { return launch_wino_fwd<1, 4, 3, 2>(fw, s); }});
                vs.push_back({"fwd Winograd F(2x2,3x3) 32x8, no output statistics", [&](hipStream_t s) { return launch_wino_fwd<1, 4, 3, 2>(fwn, s); }});
            }
        }
        bench(vs, f.out, (size_t)12 * plane, flops);      // compares sample 0's 12 planes
    }

    // ---------------- block-fused dgrad: the 4 layers of a block into its c0 = cin - 36 base channels ----------------
    {
        const int c0 = cin - 36;
        DgradBlockParams p{};
        p.n = n; p.h = h; p.w = w;
        // reuse gbuf: maps [0, c0) = gradient buffer of the base, take the 48 dY maps from buf's first planes (values only matter for timing)
        p.g = buf; p.g_ns = t * plane; p.g_cs = (int)plane; p.g_w = w;
        p.x = buf; p.out = gbuf; p.ns = t * plane; p.cs = (int)plane; p.count = c0; p.acc_from = 1 << 30; p.w_ci_off = 0;
        for (int j = 0; j < 4; ++j) {
            p.wgt[j] = wgt; p.w_cin[j] = cin; p.saved[j] = saved; p.gamma[j] = gamma; p.beta[j] = beta; p.scratch[j] = scratch;
        }
        const double bflops = 2.0 * n * plane * c0 * 12 * 9 * 4;
        std::vector<Variant> vs;
        vs.push_back({"dgrad_block<4> GP1 pipelined (library)", [&](hipStream_t s) { return launch_dgrad_block<4, 2, 3, 1, 0, 1>(p, s); }});
        vs.push_back({"dgrad_block8<4> (512 threads, 2 halves)", [&](hipStream_t s) { return launch_dgrad_block8<4>(p, s); }});
        vs.push_back({"dgrad_block<4> GP1 unpipelined", [&](hipStream_t s) { return launch_dgrad_block<4, 2, 3, 1, 0, 0>(p, s); }});
        vs.push_back({"dgrad_block<4> GP2 pipelined", [&](hipStream_t s) { return launch_dgrad_block<4, 2, 3, 2, 0, 1>(p, s); }});
        vs.push_back({"dgrad_block<4> GP2 unpipelined", [&](hipStream_t s) { return launch_dgrad_block<4, 2, 3, 2, 0, 0>(p, s); }});
        vs.push_back({"dgrad_block<4> GP1 pipelined no loads/stores", [&](hipStream_t s) { return launch_dgrad_block<4, 2, 3, 1, 3, 1>(p, s); }});
        vs.push_back({"dgrad_block<4> GP1 trivial epilogue (32+8)", [&](hipStream_t s) { return launch_dgrad_block<4, 2, 3, 1, 40, 1>(p, s); }});
        vs.push_back({"dgrad_block<4> GP1 trivial epi, no ld/st (43)", [&](hipStream_t s) { return launch_dgrad_block<4, 2, 3, 1, 43, 1>(p, s); }});
        vs.push_back({"dgrad_block<4> GP1 + weights once (47)", [&](hipStream_t s) { return launch_dgrad_block<4, 2, 3, 1, 47, 1>(p, s); }});
        vs.push_back({"dgrad_block<4> GP2 + weights once (47)", [&](hipStream_t s) { return launch_dgrad_block<4, 2, 3, 2, 47, 0>(p, s); }});
        bench(vs, p.out, (size_t)c0 * plane, bflops);
    }

    // ---------------- wgrad ----------------
    WgradParams g{};
    g.n = n; g.h = h; g.w = w; g.tiles_x = (w + kWgTileX - 1) / kWgTileX; g.tiles_y = (h + kWgTileY - 1) / kWgTileY;
    g.in = buf; g.in_ns = t * plane; g.in_cs = (int)plane; g.in_w = w; g.cin = cin;
    g.saved = saved; g.gamma = gamma; g.beta = beta;
    g.dy = gbuf + cin * plane; g.dy_ns = t * plane; g.dy_cs = (int)plane; g.dy_w = w; g.cout = 12;
    g.dw = dw;
    {
        const int groups = (cin + 15) / 16, passes = (groups + 11) / 12;
        std::vector<Variant> vs;
        auto zero = [&](hipStream_t s) { CK(hipMemsetAsync(dw, 0, (size_t)12 * cin * 9 * 4, s)); };
        vs.push_back({"wgrad taps-in-M dma 32x8", [&](hipStream_t s) { zero(s); return launch_wgrad_taps<12, IN_BNRELU>(g, s); }});
        vs.push_back({"wgrad nsplit (library)", [&](hipStream_t s) { zero(s); return launch_wgrad_nsplit(g, wscratch, s); }});
        vs.push_back({"wgrad nsplit, bf16 MFMA operands", [&](hipStream_t s) { zero(s); return launch_wgrad_nsplit(g, wscratch, s, 1); }});
        vs.push_back({"wgrad nsplit<3> no x loads", [&](hipStream_t s) { zero(s); return launch_wgrad_nsplit_ng<3, 1>(g, wscratch, passes, s); }});
        vs.push_back({"wgrad nsplit<3> no dY DMA", [&](hipStream_t s) { zero(s); return launch_wgrad_nsplit_ng<3, 2>(g, wscratch, passes, s); }});
        vs.push_back({"wgrad nsplit<3> no loads at all", [&](hipStream_t s) { zero(s); return launch_wgrad_nsplit_ng<3, 3>(g, wscratch, passes, s); }});
        vs.push_back({"wgrad nsplit<3> no loads, no barrier", [&](hipStream_t s) { zero(s); return launch_wgrad_nsplit_ng<3, 7>(g, wscratch, passes, s); }});
        bench(vs, dw, (size_t)12 * cin * 9, flops);
    }
    // ---------------- transition down, forward: BN + ReLU -> conv1x1 C -> C -> 2x2 max-pool (+ argmax codes); cin of the command line = C ----------------
    {
        const int C = cin;
        const int64_t pplane = plane / 4;
        float* tin = dev_random((size_t)n * C * plane, -1.f, 1.f, 11);
        float* tout; CK(hipMalloc(&tout, (size_t)n * C * pplane * sizeof(float)));
        uint8_t* tidx; CK(hipMalloc(&tidx, (size_t)n * C * pplane));
        float* twgt = dev_random((size_t)C * C, -0.1f, 0.1f, 12);
        float* tbias = dev_random(C, -0.1f, 0.1f, 13);
        double* tsums; CK(hipMalloc(&tsums, 2 * C * sizeof(double)));
        { std::vector<double> hsm(2 * C); for (int c = 0; c < C; ++c) { hsm[2 * c] = 0.01 * (c % 5) * cnt; hsm[2 * c + 1] = (1.0 / 3.0 + 1e-4 * (c % 5) * (c % 5)) * cnt; }
          CK(hipMemcpy(tsums, hsm.data(), hsm.size() * sizeof(double), hipMemcpyHostToDevice)); }
        double* tosums; CK(hipMalloc(&tosums, 2 * C * sizeof(double))); CK(hipMemset(tosums, 0, 2 * C * sizeof(double)));
        float* tg = dev_random(C, 0.8f, 1.2f, 14); float* tb = dev_random(C, -0.1f, 0.1f, 15);
        ConvParams q{};
        q.n = n; q.h = h; q.w = w;
        q.in = tin; q.in_ns = C * plane; q.in_cs = (int)plane; q.in_w = w; q.cin = C;
        q.in_sums = tsums; q.gamma = tg; q.beta = tb; q.running_mean = rm; q.running_var = rv; q.saved = nullptr;
        q.count = cnt; q.eps = 1e-5f; q.momentum = 0.f; q.training = 1;
        q.wgt = twgt; q.bias = tbias; q.w_cout = C; q.w_cin = C;
        q.out = tout; q.out_ns = C * pplane; q.out_cs = (int)pplane; q.out_w = w / 2; q.cout = C; q.out_sums = tosums;
        q.out_idx = tidx; q.idx_ns = C * pplane;
        printf("transition down forward: N=%d %dx%d C=%d\n", n, h, w, C);
        std::vector<Variant> vs;
        vs.push_back({"td fwd KC8 Q3 32x8 (library)", [&](hipStream_t s) { return launch_conv_dma_vec<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 2, 4, 2, 1, 4, 0>(q, s); }});
        vs.push_back({"td fwd only first chunk DMA'd (EXP 1)", [&](hipStream_t s) { return launch_conv_dma_vec<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 2, 4, 2, 1, 4, 0, 1>(q, s); }});
        vs.push_back({"td fwd no BN transform (EXP 2)", [&](hipStream_t s) { return launch_conv_dma_vec<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 2, 4, 2, 1, 4, 0, 2>(q, s); }});
        vs.push_back({"td fwd DMA not waited (EXP 4)", [&](hipStream_t s) { return launch_conv_dma_vec<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 2, 4, 2, 1, 4, 0, 4>(q, s); }});
        vs.push_back({"td fwd neither (EXP 3)", [&](hipStream_t s) { return launch_conv_dma_vec<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 2, 4, 2, 1, 4, 0, 3>(q, s); }});
        vs.push_back({"td fwd KC16 Q3 32x8", [&](hipStream_t s) { return launch_conv_dma_vec<1, 16, 3, IN_BNRELU, EPI_FWD_POOL, 2, 4, 2, 1, 4, 0>(q, s); }});
        vs.push_back({"td fwd KC8 Q3 32x8, 2 blocks per SIMD wave bound (MINW 2)", [&](hipStream_t s) { return launch_conv_dma_vec<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 2, 4, 2, 2, 4, 0>(q, s); }});
        vs.push_back({"td fwd KC8 Q3 32x4 (R 2)", [&](hipStream_t s) { return launch_conv_dma_vec<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 2, 2, 2, 1, 4, 0>(q, s); }});
        vs.push_back({"td fwd KC8 Q3 64x4 (WX 4)", [&](hipStream_t s) { return launch_conv_dma_vec<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 4, 4, 2, 1, 4, 0>(q, s); }});
        vs.push_back({"td fwd KC8 Q2 32x8", [&](hipStream_t s) { return launch_conv_dma_vec<1, 8, 2, IN_BNRELU, EPI_FWD_POOL, 2, 4, 2, 1, 4, 0>(q, s); }});
        // the coarse levels' shape (16 x 8 tiles: launch_conv_dma_auto's last branch) and its variants
        vs.push_back({"td fwd KC8 Q3 16x8 (library at levels 2-4)", [&](hipStream_t s) { return launch_conv_dma_vec<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 1, 2, 2, 1, 4, 0>(q, s); }});
        vs.push_back({"td fwd KC8 Q3 16x8 MINW 2", [&](hipStream_t s) { return launch_conv_dma_vec<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 1, 2, 2, 2, 4, 0>(q, s); }});
        vs.push_back({"td fwd KC16 Q3 16x8", [&](hipStream_t s) { return launch_conv_dma_vec<1, 16, 3, IN_BNRELU, EPI_FWD_POOL, 1, 2, 2, 1, 4, 0>(q, s); }});
        vs.push_back({"td fwd KC16 Q3 16x8 MINW 2", [&](hipStream_t s) { return launch_conv_dma_vec<1, 16, 3, IN_BNRELU, EPI_FWD_POOL, 1, 2, 2, 2, 4, 0>(q, s); }});
        vs.push_back({"td fwd KC8 Q3 16x4", [&](hipStream_t s) { return launch_conv_dma_vec<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 1, 1, 2, 1, 4, 0>(q, s); }});
        vs.push_back({"td fwd KC16 Q3 16x4 MINW 2", [&](hipStream_t s) { return launch_conv_dma_vec<1, 16, 3, IN_BNRELU, EPI_FWD_POOL, 1, 1, 2, 2, 4, 0>(q, s); }});
        vs.push_back({"td fwd KC16 Q2 16x8 MINW 2", [&](hipStream_t s) { return launch_conv_dma_vec<1, 16, 2, IN_BNRELU, EPI_FWD_POOL, 1, 2, 2, 2, 4, 0>(q, s); }});
        bench(vs, tout, (size_t)C * pplane, 2.0 * n * plane * C * C);
    }
    return 0;
}
