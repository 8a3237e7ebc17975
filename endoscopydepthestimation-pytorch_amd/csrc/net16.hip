// bf16-storage kernel family: host side (round 3).  Layout conversion and the single-convolution bricks the tests
// drive, and on top of them FCDenseNet57 (reference models.py:171-187) over bf16 level buffers in 32-channel blocks: endo_net16_fwd
// (training mode: batch statistics + running-statistics update; inference) and endo_net16_bwd (parameter gradients into the fp32
// family's flat gradient buffer).  One or two sample groups per call (a training pair's two frames, each with its own statistics).
// A separate network implementation next to the fp32 one (net.hip): same parameters, same tensors at the boundary.
#include <vector>

#include "bf16_conv_kernels.h"
#include "bf16_bwd_kernels.h"
#include "bf16_dgrad_block_kernels.h"

extern "C" int64_t endo_net_param_offset(int index);
extern "C" int64_t endo_net_bn_offset(int bn_index, int which);

// the half-storage build (net16h.hip) exports the network entry points as endo_net16h_*; the single-convolution bricks exist once (bf16)
#ifdef ENDO16_HALF
#define N16(name) endo_net16h_##name
#else
#define N16(name) endo_net16_##name
#endif

namespace endo {
inline namespace ENDO16_NS {

constexpr int k16Levels = 5, k16Layers = 4, k16Growth = 12, k16First = 48, k16New = 48;
inline int c16_down_in(int level) { return k16First + k16New * level; }
constexpr int k16Blk = 32;                                                    // channels per block of a level buffer (bf16_conv_kernels.h)
inline int c16_skip(int level) { return c16_down_in(level) + k16New; }        // channels a level hands to the up path (models.py:176-177)
// Level buffer, l < 5: [0, S) the down path (dense-block input, then its 48 new maps: the skip), [S, S + 48) the transition-up output,
// [S + 48, S + 96) the up block's new maps, S = c16_skip(l); every dense layer reads a run that starts at channel 0.  The reference
// concatenates [transition-up output, skip] (models.py:183): the up layers' parameters are indexed through (rot = 48, rot_n = S + 48).
// Bottleneck: [0, 288) input, [288, 336) new.  Padded to whole blocks.
inline int c16_level_channels(int level) {
    const int c = level < k16Levels ? 96 + c16_skip(level) : 288 + k16New;
    return (c + k16Blk - 1) / k16Blk * k16Blk;
}

// w16 / w16d: element offsets of the forward / data-gradient (transposed, flipped; -1 = none) bf16 weights
// base / koff: a dense layer's data-gradient weights below row `base` (the block's base channels) sit at k = koff + co -- the K window
// of bf16_dgrad_block_kernel -- and at k = co from there on
struct Conv16 { int64_t w, b; int cout, cin, ks, nt; int64_t w16; int rot, rot_n; int64_t w16d; int base, koff; };      // w16: element offset of the converted weights
struct Bn16 { int64_t g, b; int c; int64_t run_mean, run_var; int64_t saved; };

// the reference's module order (models.py:100-170, the order of .parameters()): offsets come from the fp32 family's own table
struct Table16 {
    Conv16 first, final_;
    Conv16 down_conv[k16Levels][k16Layers], td_conv[k16Levels], bott_conv[k16Layers], tu_conv[k16Levels], up_conv[k16Levels][k16Layers];
    Bn16 down_bn[k16Levels][k16Layers], td_bn[k16Levels], bott_bn[k16Layers], up_bn[k16Levels][k16Layers];
    std::vector<Conv16*> convs;
    int64_t w16_elems = 0, w16d_elems = 0, saved_floats = 0;
    std::vector<Conv16*> dconvs;          // the convolutions with a data gradient (all but the first)
    std::vector<Bn16*> bns;               // module order
};

static const Table16& table16() {
    static Table16* t = [] {
        Table16* tb = new Table16();
        int pi = 0, bi = 0;
        int64_t saved = 0;
        auto conv = [&](Conv16& c, int cout, int cin, int ks, bool mfma) {
            c.cout = cout; c.cin = cin; c.ks = ks; c.nt = cout <= 16 ? 1 : 3;
            c.w = endo_net_param_offset(pi++); c.b = endo_net_param_offset(pi++);
            c.w16 = -1; c.rot = 0; c.rot_n = 0; c.w16d = -1; c.base = 0; c.koff = 0;
            if (mfma && cin >= 4) {          // as a convolution over the gradient: cout' = cin (48-wide groups), K = cout
                c.w16d = tb->w16d_elems;
                const int64_t groups = (cin + 47) / 48, chunks = (cout + kBfKC - 1) / kBfKC;
                tb->w16d_elems += chunks * groups * ks * ks * 3 * 16 * 32;
                tb->dconvs.push_back(&c);
            }
            if (mfma) {
                const int cin_k = cin < 4 ? 4 : cin;          // the first convolution's 3 input channels travel as 4 (+ 4 zero) of an 8-channel record
                c.w16 = tb->w16_elems;
                const int64_t groups = (cout + c.nt * 16 - 1) / (c.nt * 16), chunks = (cin_k + kBfKC - 1) / kBfKC;
                tb->w16_elems += chunks * groups * ks * ks * c.nt * 16 * 32;
                tb->convs.push_back(&c);
            }
        };
        auto bn = [&](Bn16& b, int c) {
            b.c = c;
            b.g = endo_net_param_offset(pi++); b.b = endo_net_param_offset(pi++);
            b.run_mean = endo_net_bn_offset(bi, 0); b.run_var = endo_net_bn_offset(bi, 1); ++bi;
            b.saved = saved; saved += 2 * c;
            tb->bns.push_back(&b);
        };
        const int koffs[k16Layers] = {0, 12, 8, 20};          // 12 j - 8 u0_j, u0 = {0, 0, 2, 2} (bf16_dgrad_block_kernels.h)
        conv(tb->first, k16First, 3, 3, true);
        for (int l = 0; l < k16Levels; ++l)
            for (int j = 0; j < k16Layers; ++j) {
                bn(tb->down_bn[l][j], c16_down_in(l) + k16Growth * j); conv(tb->down_conv[l][j], k16Growth, c16_down_in(l) + k16Growth * j, 3, true);
                tb->down_conv[l][j].base = c16_down_in(l); tb->down_conv[l][j].koff = koffs[j];
            }
        for (int l = 0; l < k16Levels; ++l) { bn(tb->td_bn[l], c16_down_in(l) + k16New); conv(tb->td_conv[l], c16_down_in(l) + k16New, c16_down_in(l) + k16New, 1, true); }
        for (int j = 0; j < k16Layers; ++j) {
            bn(tb->bott_bn[j], 288 + k16Growth * j); conv(tb->bott_conv[j], k16Growth, 288 + k16Growth * j, 3, true);
            tb->bott_conv[j].base = 288; tb->bott_conv[j].koff = koffs[j];
        }
        for (int i = 0; i < k16Levels; ++i) conv(tb->tu_conv[i], k16New, k16New, 3, true);
        for (int i = 0; i < k16Levels; ++i) {
            const int l = k16Levels - 1 - i;
            for (int j = 0; j < k16Layers; ++j) {
                const int cin = 96 + c16_down_in(l) + k16Growth * j;
                bn(tb->up_bn[i][j], cin); conv(tb->up_conv[i][j], k16Growth, cin, 3, true);
                tb->up_conv[i][j].rot = k16New; tb->up_conv[i][j].rot_n = c16_skip(l) + k16New;
                tb->up_conv[i][j].base = c16_skip(l) + k16New; tb->up_conv[i][j].koff = koffs[j];
            }
        }
        conv(tb->final_, 1, 192, 1, false);
        tb->final_.rot = k16New; tb->final_.rot_n = c16_skip(0) + k16New;
        tb->saved_floats = saved;
        return tb;
    }();
    return *t;
}

// every MFMA convolution's weights, fp32 -> the kernel's bf16 layout, in ONE launch (the parameters change every step)
struct W16Table {
    int layers;
    int64_t start[64];          // prefix sum of output elements
    int64_t w[63], out[63];
    int cout[63], cin[63], cin_k[63], ks[63], nt[63], rot[63], rot_n[63], base[63], koff[63];
    int dgrad;          // 1: data-gradient form: row = input channel of the layer, k = its cout, taps flipped
};

__global__ void __launch_bounds__(256) bf16_all_weights_kernel(const W16Table t, const float* __restrict__ params, uint16_t* __restrict__ w16) {
    const int64_t total = t.start[t.layers];
    for (int64_t item = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; item < total; item += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        int l = 0;
        while (item >= t.start[l + 1]) ++l;
        const int64_t e = item - t.start[l];
        const int taps = t.ks[l] * t.ks[l], nt = t.nt[l];
        const int ngroups = (t.cout[l] + nt * 16 - 1) / (nt * 16);
        const int k = e & 31, co16 = (e >> 5) & 15;
        int64_t rest = e >> 9;
        const int tt = rest % nt; rest /= nt;
        const int tap = rest % taps; rest /= taps;
        const int grp = rest % ngroups;
        const int chunk = rest / ngroups;
        const int co = (grp * nt + tt) * 16 + co16;
        int ci = chunk * kBfKC + k;
        // rows of a dense block's base channels are copied into LDS as they lie (bf16_dgrad_block_kernel): position k holds slot k / 8
        // of the row, which is channel part (k / 8) XOR (row >> 1) & 3
        if (t.dgrad && co < t.base[l]) ci = (((k >> 3) ^ ((co16 >> 1) & 3)) << 3) | (k & 7);
        float v = 0.f;
        if (t.dgrad) {          // here cout[l] / cin[l] are the CONVOLUTION's: rows = the layer's input channels, k = the layer's couts
            const int pco = co < t.rot_n[l] ? (co + t.rot[l] < t.rot_n[l] ? co + t.rot[l] : co + t.rot[l] - t.rot_n[l]) : co;
            const int kk = ci - (co < t.base[l] ? t.koff[l] : 0);          // the layer's cout at this k
            if (co < t.cout[l] && kk >= 0 && kk < t.cin[l]) v = params[t.w[l] + (static_cast<int64_t>(kk) * t.cout[l] + pco) * taps + (taps - 1 - tap)];
        } else {
            const int pci = ci < t.rot_n[l] ? (ci + t.rot[l] < t.rot_n[l] ? ci + t.rot[l] : ci + t.rot[l] - t.rot_n[l]) : ci;
            if (co < t.cout[l] && ci < t.cin[l]) v = params[t.w[l] + (static_cast<int64_t>(co) * t.cin[l] + pci) * taps + tap];
        }
        w16[t.out[l] + e] = static_cast<uint16_t>(pack_s16x2(v, 0.f) & 0xffffu);
    }
}

// x: fp32 [n][3][h][w] -> bf16 [n][h][w][8] (channels 3..7 zero): the first convolution's input record
__global__ void __launch_bounds__(256) bf16_pack_input_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, int plane) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        const float* xp = x + static_cast<int64_t>(n) * 3 * plane + i;
        const unsigned a = pack_s16x2(xp[0], xp[plane]), b = pack_s16x2(xp[2 * static_cast<int64_t>(plane)], 0.f);
        *reinterpret_cast<u32x4_t*>(out + (static_cast<int64_t>(n) * plane + i) * 8) = u32x4_t{a, b, 0u, 0u};
    }
}

// final 1x1 convolution 192 -> 1 and |.| (reference models.py:167, 186) over the level-0 buffer ([6 blocks][plane][32]): 8 lanes per
// pixel, 24 channels (three 8-channel units) each; the weight of buffer channel c is w[(c + rot) % rot_n] for c < rot_n
__global__ void __launch_bounds__(256) bf16_final_fwd_kernel(const uint16_t* __restrict__ u, int64_t ns, int plane, const float* __restrict__ w,
                                                             const float* __restrict__ bias, int rot, int rot_n, float* __restrict__ pre,
                                                             float* __restrict__ out) {
    const int n = blockIdx.y;
    const int sub = threadIdx.x & 7;
    float wv[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) {
        const int c = sub * 24 + k;
        wv[k] = w[c < rot_n ? (c + rot < rot_n ? c + rot : c + rot - rot_n) : c];
    }
    for (int px = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; px < plane; px += (gridDim.x * blockDim.x) >> 3) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c0 = sub * 24 + 8 * j;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(u + n * ns + (static_cast<int64_t>(c0 >> 5) * plane + px) * k16Blk + (c0 & 31));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc = fmaf(s16_lo(v[k]), wv[8 * j + 2 * k], acc);
                acc = fmaf(s16_hi(v[k]), wv[8 * j + 2 * k + 1], acc);
            }
        }
        acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
        if (sub == 0) {
            const float z = acc + bias[0];
            pre[static_cast<int64_t>(n) * plane + px] = z;
            out[static_cast<int64_t>(n) * plane + px] = fabsf(z);
        }
    }
}

// fp32 NCHW [n][c][h][w] -> channels [oc0, oc0 + c) of a bf16 buffer [n][t / blk][h][w][blk] (blk = t: channels-last)
__global__ void __launch_bounds__(256) bf16_pack_nhwc_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, int c, int plane, int t, int blk,
                                                             int oc0) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane * c; i += gridDim.x * blockDim.x) {
        const int ch = i % c, px = i / c;
        const float v = x[(static_cast<int64_t>(n) * c + ch) * plane + px];
        const int ca = oc0 + ch, cb = ca / blk;
        out[static_cast<int64_t>(n) * plane * t + (static_cast<int64_t>(cb) * plane + px) * blk + (ca - cb * blk)] =
            static_cast<uint16_t>(pack_s16x2(v, 0.f) & 0xffffu);
    }
}

__global__ void __launch_bounds__(256) bf16_unpack_nhwc_kernel(const uint16_t* __restrict__ in, float* __restrict__ x, int c, int plane, int t, int blk,
                                                               int ic0) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane * c; i += gridDim.x * blockDim.x) {
        const int ch = i % c, px = i / c;
        const int ca = ic0 + ch, cb = ca / blk;
        x[(static_cast<int64_t>(n) * c + ch) * plane + px] =
            s16_lo(in[static_cast<int64_t>(n) * plane * t + (static_cast<int64_t>(cb) * plane + px) * blk + (ca - cb * blk)]);
    }
}

}  // inline namespace
}  // namespace endo

using namespace endo;

#ifdef ENDO16_HALF
#define B16(name) endo_f16_##name
#else
#define B16(name) endo_bf16_##name
#endif
// blk: channels per block of the buffer ([n][t / blk][h][w][blk]); 0 or t = plain channels-last
extern "C" int B16(pack_nhwc)(const float* x, void* out, int n, int c, int h, int w, int t, int blk, int oc0, void* stream) {
    if (blk <= 0) blk = t;
    if (!x || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0 || oc0 < 0 || oc0 + c > t || t % blk) return ENDO_E_BADARG;
    bf16_pack_nhwc_kernel<<<dim3(256, n), 256, 0, static_cast<hipStream_t>(stream)>>>(x, static_cast<uint16_t*>(out), c, h * w, t, blk, oc0);
    ENDO_LAUNCH_CHECK();
    return 0;
}

extern "C" int B16(unpack_nhwc)(const void* in, float* x, int n, int c, int h, int w, int t, int blk, int ic0, void* stream) {
    if (blk <= 0) blk = t;
    if (!x || !in || n <= 0 || c <= 0 || h <= 0 || w <= 0 || ic0 < 0 || ic0 + c > t || t % blk) return ENDO_E_BADARG;
    bf16_unpack_nhwc_kernel<<<dim3(256, n), 256, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const uint16_t*>(in), x, c, h * w, t, blk, ic0);
    ENDO_LAUNCH_CHECK();
    return 0;
}
#ifndef ENDO16_HALF
extern "C" int64_t endo_bf16_conv_weight_elems(int cout, int cin, int ks) {
    if (cout <= 0 || cin <= 0 || (ks != 1 && ks != 3)) return -1;
    const int nt = cout <= 16 ? 1 : 3;
    const int64_t groups = (cout + nt * 16 - 1) / (nt * 16), chunks = (cin + kBfKC - 1) / kBfKC;
    return chunks * groups * ks * ks * nt * 16 * 32;
}

extern "C" int endo_bf16_conv_weights(const float* w, int cout, int cin, int ks, void* out, void* stream) {
    if (!w || !out || endo_bf16_conv_weight_elems(cout, cin, ks) < 0) return ENDO_E_BADARG;
    const int nt = cout <= 16 ? 1 : 3;
    bf16_conv_weights_kernel<<<256, 256, 0, static_cast<hipStream_t>(stream)>>>(w, cout, cin, ks, nt, static_cast<uint16_t*>(out), 0, 0);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// One convolution over bf16 buffers (reference models.py:22-25 / 73-74: [BN -> ReLU ->] [nearest x2 ->] conv KS x KS + bias).
// in: [n][in_t / in_blk][in_h][in_w][in_blk] bf16, channels [ic0, ic0 + cin); out: [n][out_t / out_blk][h][w][out_blk] bf16, channels
// [oc0, oc0 + cout) (blk 0 = t: channels-last; cout % 4 == 0, cin % 4 == 0, ic0 % 8 == 0, in_blk % 8 == 0, oc0 % 4 == 0, out_blk % 4 == 0); bn: [cin][2] (scale, shift) or null; wgt from endo_bf16_conv_weights;
// out_sums: [cout][2] fp64 (sum, sum^2 of the stored values, ACCUMULATED) or null; ups: input is (h / 2) x (w / 2), nearest x2.
extern "C" int endo_bf16_conv(const void* in, int in_t, int in_blk, int ic0, int cin, const float* bn, const void* wgt, const float* bias, void* out,
                              int out_t, int out_blk, int oc0, int cout, double* out_sums, int n, int h, int w, int ks, int ups, void* stream_) {
    if (!in || !wgt || !out || n <= 0 || h <= 0 || w <= 0 || (ks != 1 && ks != 3) || cin <= 0 || cout <= 0 || (cin & 3) || (cout & 3) || (ic0 & 7) ||
        (in_t & 7) || (oc0 & 3) || (out_t & 3) || ic0 + cin > in_t || oc0 + cout > out_t || (ups && ((h | w) & 1)))
        return ENDO_E_BADARG;
    Conv16Params p{};
    p.n = n; p.h = h; p.w = w;
    p.in_h = ups ? h / 2 : h; p.in_w = ups ? w / 2 : w;
    p.in_blk = in_blk; p.out_blk = out_blk;
    p.in = static_cast<const uint16_t*>(in); p.in_t = in_t; p.in_ns = static_cast<int64_t>(p.in_h) * p.in_w * in_t; p.ic0 = ic0; p.cin = cin;
    p.bn = bn; p.wgt = static_cast<const uint16_t*>(wgt); p.bias = bias;
    p.out = static_cast<uint16_t*>(out); p.out_t = out_t; p.out_ns = static_cast<int64_t>(h) * w * out_t; p.oc0 = oc0; p.cout = cout;
    p.out_sums = out_sums; p.ups = ups;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (ks == 3) return cout <= 16 ? launch_bf16_conv<3, 1, 0, 8, 4>(p, stream) : launch_bf16_conv<3, 3, 0, 8, 2>(p, stream);
    return cout <= 16 ? launch_bf16_conv<1, 1, 0, 8, 4>(p, stream) : launch_bf16_conv<1, 3, 0, 8, 4>(p, stream);
}


#endif  // !ENDO16_HALF

// ---------------------------------------------------------------------------------------------
// FCDenseNet57 forward over bf16 level buffers (reference models.py:171-187)
// ---------------------------------------------------------------------------------------------
struct endo_net16 {
    int n, h, w;               // n: all samples of a call = groups x gn
    int gn, groups;            // sample groups: each has its own BatchNorm batch statistics (bf16_conv_kernels.h)
    int64_t gs_saved, gs_sums; // floats / doubles between the groups' (mean, rstd) tables and forward sums (the backward sums: gs_saved doubles)
    struct Level { int h, w, t; int64_t plane; int64_t act; int64_t sums; } lv[k16Levels + 1];      // act: bytes, sums: doubles (from sums_off)
    int64_t in_off;            // bytes: the packed input [n][h][w][8] bf16
    int64_t idx_off[k16Levels];
    int64_t pre_off;           // bytes: final pre-activation fp32
    int64_t saved_off;         // bytes: BN (mean, rstd) fp32
    int64_t sums_off, sums_bytes;
    int64_t w16_off;           // bytes
    int64_t tape_bytes;
    // backward workspace (endo_net16_bwd): gradient buffers with the geometry of the level buffers, the deferred BatchNorm terms P, Q
    // per level channel, the BatchNorm backward sums per layer, the data-gradient weights, the weight-gradient partials
    int64_t ws_d[k16Levels + 1];          // bytes
    int64_t ws_pq[k16Levels + 1];         // bytes: floats [2][t]
    int64_t ws_bnsums;                    // bytes: doubles, layer at its `saved` offset; kBnSlots copies, bn_slot_stride doubles apart
    int64_t bn_slot_stride;
    int64_t ws_gsum[k16Levels + 1];       // bytes: doubles [t][2], pixel sums of the total gradient per level channel (bf16_prep_dy_kernel)
    int64_t ws_zero_begin, ws_zero_end;   // everything but the level-0 gradient buffer starts at zero
    int64_t ws_w16d;                      // bytes
    int64_t ws_partial;                   // bytes
    int64_t ws_partial_floats;            // capacity of the partial buffer (launch_bf16_wgrad checks every launch against it)
    int64_t ws_gscale;                    // bytes: {S, 1 / S} of the stored gradients (half storage)
    int64_t ws_bytes;
    // the weight gradients read only finished tensors (forward activations, a prepared gradient range) and nothing waits for them but
    // the optimizer: they run on a side stream, forked after every prep_dy and joined once at the end (as in the fp32 family)
    hipStream_t wstream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int use_wstream = 1;
};

extern "C" int N16(create)(endo_net16** out, int n_per_group, int h, int w, int groups) {
    if (!out || n_per_group <= 0 || h <= 0 || w <= 0 || (h % 32) || (w % 32) || groups < 1 || groups > 2) return ENDO_E_BADARG;
    const int n = n_per_group * groups;
    const Table16& tb = table16();
    endo_net16* net = new (std::nothrow) endo_net16();
    if (!net) return ENDO_E_BADARG;
    net->n = n; net->h = h; net->w = w; net->gn = n_per_group; net->groups = groups;
    auto align = [](int64_t v) { return (v + 255) / 256 * 256; };
    int64_t off = 0, sums = 0;
    for (int l = 0; l <= k16Levels; ++l) {
        auto& lv = net->lv[l];
        lv.h = h >> l; lv.w = w >> l; lv.t = c16_level_channels(l); lv.plane = static_cast<int64_t>(lv.h) * lv.w;
        lv.act = off; off += align(static_cast<int64_t>(n) * lv.plane * lv.t * 2);
        lv.sums = sums; sums += 2 * lv.t;
    }
    net->in_off = off; off += align(static_cast<int64_t>(n) * net->lv[0].plane * 16);
    for (int l = 0; l < k16Levels; ++l) { net->idx_off[l] = off; off += align(static_cast<int64_t>(n) * net->lv[l + 1].plane * (c16_down_in(l) + k16New)); }
    net->pre_off = off; off += align(static_cast<int64_t>(n) * net->lv[0].plane * 4);
    net->gs_saved = tb.saved_floats; net->gs_sums = sums;
    net->saved_off = off; off += align(tb.saved_floats * 4 * groups);
    net->sums_off = off; net->sums_bytes = sums * 8 * groups; off += align(net->sums_bytes);
    net->w16_off = off; off += align(tb.w16_elems * 2);
    net->tape_bytes = off;
    {
        int64_t o = 0;
        net->ws_d[0] = o; o += align(static_cast<int64_t>(n) * net->lv[0].plane * net->lv[0].t * 2);
        net->ws_zero_begin = o;
        for (int l = 1; l <= k16Levels; ++l) { net->ws_d[l] = o; o += align(static_cast<int64_t>(n) * net->lv[l].plane * net->lv[l].t * 2); }
        for (int l = 0; l <= k16Levels; ++l) { net->ws_pq[l] = o; o += align(static_cast<int64_t>(net->lv[l].t) * 2 * 4 * groups); }
        net->bn_slot_stride = tb.saved_floats * groups;          // kBnSlots copies of the BatchNorm-backward sums (common.h), this many doubles apart
        net->ws_bnsums = o; o += align(tb.saved_floats * 8 * groups * kBnSlots);
        for (int l = 0; l <= k16Levels; ++l) { net->ws_gsum[l] = o; o += align(static_cast<int64_t>(net->lv[l].t) * 2 * 8); }
        net->ws_zero_end = o;
        net->ws_w16d = o; o += align(tb.w16d_elems * 2);
        int64_t need = 0;
        auto wg = [&](int level, int cin, int cout, int ks) {
            Wgrad16Params q{};
            q.n = n; q.h = net->lv[level].h; q.w = net->lv[level].w; q.cin = cin; q.cout = cout;
            const int64_t f = bf16_wgrad_partial_floats(cin, cout, ks, bf16_wgrad_blocks(q, ks));
            need = f > need ? f : need;
        };
        // EVERY weight-gradient launch of the backward pass: blocks x co_groups x 144 x ci_pad is not monotonic in cin (the block count
        // drops when cin crosses a multiple of 64), so the widest layer of a level is not its largest partial buffer
        wg(0, 4, k16First, 3);
        for (int l = 0; l < k16Levels; ++l) {
            for (int j = 0; j < k16Layers; ++j) {
                wg(l, tb.down_conv[l][j].cin, k16Growth, 3);
                wg(l, tb.up_conv[k16Levels - 1 - l][j].cin, k16Growth, 3);
            }
            wg(l, tb.td_conv[l].cin, tb.td_conv[l].cout, 1);                 // transition down
            wg(l, k16New, k16New, 3);                                        // transition up (output grid of level l)
        }
        for (int j = 0; j < k16Layers; ++j) wg(k16Levels, tb.bott_conv[j].cin, k16Growth, 3);
        net->ws_partial_floats = need;
        net->ws_partial = o; o += align(need * 4);
        net->ws_gscale = o; o += 256;
        net->ws_bytes = o;
    }
    *out = net;
    return 0;
}
extern "C" void N16(destroy)(endo_net16* net) {
    if (!net) return;
    if (net->ev_fork) (void)hipEventDestroy(net->ev_fork);
    if (net->ev_join) (void)hipEventDestroy(net->ev_join);
    if (net->wstream) (void)hipStreamDestroy(net->wstream);
    delete net;
}
// 1 (default): endo_net16_bwd issues the weight gradients on a side stream it owns (forked after every prep_dy, joined before it returns);
// 0: everything in line on the caller's stream (what a caller capturing the step into a graph on one stream wants)
extern "C" int N16(set_wgrad_overlap)(endo_net16* net, int on) {
    if (!net) return ENDO_E_BADARG;
    net->use_wstream = on < 0 ? 0 : (on > 2 ? 2 : on);          // 2: one fork per dense block (the four weight gradients behind the block's last prep_dy)
    return 0;
}
extern "C" int64_t N16(tape_bytes)(const endo_net16* net) { return net ? net->tape_bytes : 0; }
extern "C" int64_t N16(bwd_workspace_bytes)(const endo_net16* net) { return net ? net->ws_bytes : 0; }
// Where things are (tests read the forward pass's decisions and the gradient buffers): byte offsets into the tape -- what 0: final
// pre-activation (fp32), 1: (mean, rstd) of BatchNorm layer `index` in module order (fp32), 2: max-pool codes of transition down
// `index`, 3: level buffer `index` -- or into the backward workspace -- 4: gradient buffer of level `index`; 5: channels of level
// buffer `index` (not an offset); 7: bytes between two sample groups' (mean, rstd) tables.  -1 for anything else.
extern "C" int64_t N16(offset)(const endo_net16* net, int what, int index) {
    if (!net || index < 0) return -1;
    const Table16& tb = table16();
    switch (what) {
        case 0: return net->pre_off;
        case 1: return index < static_cast<int>(tb.bns.size()) ? net->saved_off + tb.bns[index]->saved * 4 : -1;
        case 2: return index < k16Levels ? net->idx_off[index] : -1;
        case 3: return index <= k16Levels ? net->lv[index].act : -1;
        case 4: return index <= k16Levels ? net->ws_d[index] : -1;
        case 5: return index <= k16Levels ? net->lv[index].t : -1;
        case 6: return index <= k16Levels ? net->ws_pq[index] : -1;          // workspace: deferred BatchNorm terms of the level, fp32 [P[t]][Q[t]] per group
        case 7: return net->gs_saved * 4;          // not an offset: bytes from one sample group's saved (mean, rstd) table to the next
        default: return -1;
    }
}

namespace {

struct Ctx16 {
    const endo_net16* net;
    const float* params;
    float* bn_running;
    char* tape;
    int training;
    hipStream_t stream;
    float* grads = nullptr;          // backward: parameter gradients (accumulated), workspace
    char* ws = nullptr;
    uint16_t* dbuf(int level) const { return reinterpret_cast<uint16_t*>(ws + net->ws_d[level]); }
    float* pq_p(int level) const { return reinterpret_cast<float*>(ws + net->ws_pq[level]); }
    float* pq_q(int level) const { return pq_p(level) + net->lv[level].t; }
    double* gsum(int level) const { return reinterpret_cast<double*>(ws + net->ws_gsum[level]); }
    double* bnsums(const Bn16& b) const { return reinterpret_cast<double*>(ws + net->ws_bnsums) + b.saved; }
    const uint16_t* w16d(const Conv16& c) const { return reinterpret_cast<const uint16_t*>(ws + net->ws_w16d) + c.w16d; }
    float* partial() const { return reinterpret_cast<float*>(ws + net->ws_partial); }
#ifdef ENDO16_HALF
    const float* gscale() const { return reinterpret_cast<const float*>(ws + net->ws_gscale); }
#else
    const float* gscale() const { return nullptr; }
#endif
    // the side stream, after everything issued on `stream` so far
    int fork_wgrad(hipStream_t& side) const {
        const bool on = net->wstream && net->use_wstream;
        side = on ? net->wstream : stream;
        if (on) {
            ENDO_CHECK(hipEventRecord(net->ev_fork, stream));
            ENDO_CHECK(hipStreamWaitEvent(net->wstream, net->ev_fork, 0));
        }
        return 0;
    }
    uint16_t* act(int level) const { return reinterpret_cast<uint16_t*>(tape + net->lv[level].act); }
    double* sums(int level) const { return reinterpret_cast<double*>(tape + net->sums_off) + net->lv[level].sums; }
    const uint16_t* w16(const Conv16& c) const { return reinterpret_cast<const uint16_t*>(tape + net->w16_off) + c.w16; }
    float* saved(const Bn16& b) const { return reinterpret_cast<float*>(tape + net->saved_off) + b.saved; }
};

void fill_io(const Ctx16& c, Conv16Params& p, int in_level, int ic0, int cin, int out_level, int oc0, const Conv16& cv) {
    const auto& li = c.net->lv[in_level];
    const auto& lo = c.net->lv[out_level];
    p.n = c.net->n;
    p.in = c.act(in_level); p.in_t = li.t; p.in_blk = k16Blk; p.in_h = li.h; p.in_w = li.w; p.in_ns = li.plane * li.t; p.ic0 = ic0; p.cin = cin;
    p.wgt = c.w16(cv); p.bias = c.params + cv.b; p.rot = cv.rot; p.rot_n = cv.rot_n;
    p.out = c.act(out_level); p.out_t = lo.t; p.out_blk = k16Blk; p.out_ns = lo.plane * lo.t; p.oc0 = oc0; p.cout = cv.cout;
    p.out_sums = c.training ? c.sums(out_level) + 2 * oc0 : nullptr;
    p.group_n = c.net->groups > 1 ? c.net->gn : 0; p.gs_in_sums = c.net->gs_sums; p.gs_out_sums = c.net->gs_sums; p.gs_saved = c.net->gs_saved;
}

void fill_bn(const Ctx16& c, Conv16Params& p, const Bn16& b, int level, int ic0) {
    const auto& lv = c.net->lv[level];
    p.use_stats = 1;
    p.in_sums = c.sums(level) + 2 * ic0;
    p.gamma = c.params + b.g; p.beta = c.params + b.b;
    p.running_mean = c.bn_running + b.run_mean; p.running_var = c.bn_running + b.run_var;
    p.saved = c.saved(b);
    p.count = static_cast<double>(c.net->gn) * lv.plane;
    p.eps = 1.0e-5f; p.momentum = 0.1f; p.training = c.training;
}

// dense layer: BN -> ReLU -> conv3x3 -> +12 channels (models.py:19-28, 44-52)
int dense16(const Ctx16& c, int level, int ic0, int oc0, const Bn16& b, const Conv16& cv) {
    Conv16Params p{};
    fill_io(c, p, level, ic0, cv.cin, level, oc0, cv);
    p.h = c.net->lv[level].h; p.w = c.net->lv[level].w;
    fill_bn(c, p, b, level, ic0);
    const double px = static_cast<double>(c.net->n) * c.net->lv[level].plane;
    ProfScope prof(kProfConv3x3Dense, c.stream, 2.0 * px * cv.cin * cv.cout * 9, 2.0 * px * (cv.cin + cv.cout));
    return launch_bf16_conv<3, 1, 0, 8, 4>(p, c.stream);
}

// transition down: BN -> ReLU -> conv1x1 -> maxpool2 into the next level (models.py:56-67)
int td16(const Ctx16& c, int level, const Bn16& b, const Conv16& cv) {
    const int next = level + 1;
    Conv16Params p{};
    fill_io(c, p, level, 0, cv.cin, next, 0, cv);
    p.h = c.net->lv[level].h; p.w = c.net->lv[level].w;
    fill_bn(c, p, b, level, 0);
    p.out_idx = reinterpret_cast<uint8_t*>(c.tape + c.net->idx_off[level]);
    const double px = static_cast<double>(c.net->n) * c.net->lv[level].plane;
    ProfScope prof(kProfConv1x1Pool, c.stream, 2.0 * px * cv.cin * cv.cout, 2.0 * px * cv.cin + 0.75 * px * cv.cout);
    return launch_bf16_conv<1, 3, 1, 8, 4>(p, c.stream);
}

// transition up: nearest x2 -> conv3x3 48 -> 48 into channels [S, S + 48) of the finer level (models.py:70-80)
int tu16(const Ctx16& c, int level, int src_level, int src_c0, const Conv16& cv) {
    Conv16Params p{};
    fill_io(c, p, src_level, src_c0, cv.cin, level, c16_skip(level), cv);
    p.h = c.net->lv[level].h; p.w = c.net->lv[level].w;
    p.ups = 1;
    const double px = static_cast<double>(c.net->n) * c.net->lv[level].plane;
    ProfScope prof(kProfConv3x3Up, c.stream, 2.0 * px * cv.cin * cv.cout * 9, 2.0 * px * (cv.cin / 4.0 + cv.cout));
    return launch_bf16_conv<3, 3, 0, 8, 2>(p, c.stream);
}

}  // namespace

// x: fp32 [n][3][H][W] (already multiplied by the boundary, train.py:272-273; rounded to bf16 on the way in); out: fp32 [n][1][H][W] >= 0.
// training != 0: batch statistics + running-statistics update (momentum 0.1, eps 1e-5); 0: running statistics.  tape:
// endo_net16_tape_bytes() bytes, 256-byte aligned.  H, W multiples of 32.
extern "C" int N16(fwd)(endo_net16* net, const float* params, float* bn_running, const float* x, float* out, void* tape_, int training,
                              void* stream_) {
    if (!net || !params || !bn_running || !x || !out || !tape_) return ENDO_E_BADARG;
    const Table16& tb = table16();
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Ctx16 c{net, params, bn_running, static_cast<char*>(tape_), training, stream};
    ENDO_CHECK(hipMemsetAsync(c.tape + net->sums_off, 0, net->sums_bytes, stream));
    {   // weights of all 56 MFMA convolutions -> bf16
        W16Table t{};
        t.layers = static_cast<int>(tb.convs.size());
        if (t.layers > 63) return ENDO_E_BADARG;
        int64_t start = 0;
        for (int l = 0; l < t.layers; ++l) {
            const Conv16& cv = *tb.convs[l];
            const int cin_k = cv.cin < 4 ? 4 : cv.cin;
            const int64_t groups = (cv.cout + cv.nt * 16 - 1) / (cv.nt * 16), chunks = (cin_k + kBfKC - 1) / kBfKC;
            t.start[l] = start; t.w[l] = cv.w; t.out[l] = cv.w16; t.cout[l] = cv.cout; t.cin[l] = cv.cin; t.cin_k[l] = cin_k; t.ks[l] = cv.ks; t.nt[l] = cv.nt; t.rot[l] = cv.rot; t.rot_n[l] = cv.rot_n;
            start += chunks * groups * cv.ks * cv.ks * cv.nt * 16 * 32;
        }
        t.start[t.layers] = start;
        bf16_all_weights_kernel<<<1024, 256, 0, stream>>>(t, params, reinterpret_cast<uint16_t*>(c.tape + net->w16_off));
        ENDO_LAUNCH_CHECK();
    }
    int rc;
    {   // first conv 3 -> 48 into level-0 channels [48, 96)
        uint16_t* xin = reinterpret_cast<uint16_t*>(c.tape + net->in_off);
        bf16_pack_input_kernel<<<dim3(256, net->n), 256, 0, stream>>>(x, xin, static_cast<int>(net->lv[0].plane));
        ENDO_LAUNCH_CHECK();
        Conv16Params p{};
        p.n = net->n; p.h = net->h; p.w = net->w;
        p.in = xin; p.in_t = 8; p.in_h = net->h; p.in_w = net->w; p.in_ns = net->lv[0].plane * 8; p.ic0 = 0; p.cin = 4;
        p.wgt = c.w16(tb.first); p.bias = params + tb.first.b;
        p.out = c.act(0); p.out_t = net->lv[0].t; p.out_blk = k16Blk; p.out_ns = net->lv[0].plane * net->lv[0].t; p.oc0 = 0; p.cout = k16First;
        p.out_sums = training ? c.sums(0) : nullptr;
        p.group_n = net->groups > 1 ? net->gn : 0; p.gs_out_sums = net->gs_sums;
        rc = launch_bf16_conv<3, 3, 0, 8, 2>(p, stream);
        if (rc) return rc;
    }
    for (int l = 0; l < k16Levels; ++l) {
        for (int j = 0; j < k16Layers; ++j) {
            rc = dense16(c, l, 0, c16_down_in(l) + k16Growth * j, tb.down_bn[l][j], tb.down_conv[l][j]);
            if (rc) return rc;
        }
        rc = td16(c, l, tb.td_bn[l], tb.td_conv[l]);
        if (rc) return rc;
    }
    for (int j = 0; j < k16Layers; ++j) {
        rc = dense16(c, k16Levels, 0, 288 + k16Growth * j, tb.bott_bn[j], tb.bott_conv[j]);
        if (rc) return rc;
    }
    for (int i = 0; i < k16Levels; ++i) {
        const int l = k16Levels - 1 - i, src = l + 1;
        const int src_c0 = (i == 0) ? 288 : c16_skip(src) + k16New;
        rc = tu16(c, l, src, src_c0, tb.tu_conv[i]);
        if (rc) return rc;
        for (int j = 0; j < k16Layers; ++j) {
            rc = dense16(c, l, 0, c16_skip(l) + k16New + k16Growth * j, tb.up_bn[i][j], tb.up_conv[i][j]);
            if (rc) return rc;
        }
    }
    {
        const auto& lv = net->lv[0];
        int bx = static_cast<int>((lv.plane * 8 + 255) / 256);
        bx = bx > 2048 ? 2048 : bx;
        bf16_final_fwd_kernel<<<dim3(bx, net->n), 256, 0, stream>>>(c.act(0), lv.plane * lv.t, static_cast<int>(lv.plane), params + tb.final_.w,
                                                                     params + tb.final_.b, tb.final_.rot, tb.final_.rot_n, reinterpret_cast<float*>(c.tape + net->pre_off), out);
        ENDO_LAUNCH_CHECK();
    }
    return 0;
}


// ---------------------------------------------------------------------------------------------
// FCDenseNet57 backward over bf16 level buffers
// ---------------------------------------------------------------------------------------------
namespace {

int prep_dy16(const Ctx16& c, int level, int c0, int count, float* bias_grad) {
    const auto& lv = c.net->lv[level];
    const int ppi = 256 / (count / 4);
    int bx = static_cast<int>((lv.plane + ppi - 1) / ppi);
    bx = bx > 1024 ? 1024 : bx;
    const dim3 grid = c.training ? dim3(bx, c.net->n) : dim3(1, 1);          // inference mode: only the bias gradient
    bf16_prep_dy_kernel<<<grid, 256, 0, c.stream>>>(c.dbuf(level), c.act(level), lv.plane * lv.t, static_cast<int>(lv.plane), k16Blk, c0, count,
                                                                  c.pq_p(level), c.pq_q(level), bias_grad, c.gsum(level), c.training,
                                                                  c.net->groups > 1 ? c.net->gn : 0, 2 * lv.t, c.gscale());
    ENDO_LAUNCH_CHECK();
    return 0;
}

// channels [first, first + count) of the layer's input (count < 0: all of them)
int bn_finalize16(const Ctx16& c, const Bn16& b, const Conv16& cv, int level, int first = 0, int count = -1) {
    const auto& lv = c.net->lv[level];
    if (count < 0) count = b.c - first;
    if (count == 0) return 0;
    bf16_bn_finalize_kernel<<<dim3((count + 127) / 128, c.net->groups), 128, 0, c.stream>>>(
        c.bnsums(b), c.saved(b), c.params + b.g, c.grads + b.g, c.grads + b.b, c.pq_p(level), c.pq_q(level), c.gsum(level), first, count, cv.rot, cv.rot_n,
        static_cast<double>(c.net->gn) * lv.plane, c.training, c.net->gs_saved, c.net->gs_saved, 2 * lv.t, c.gscale(), c.net->bn_slot_stride);
    ENDO_LAUNCH_CHECK();
    return 0;
}

void fill_wgrad_a(const Ctx16& c, Wgrad16Params& p, int level, int ac0, int cin, const Bn16* b, const Conv16& cv) {
    const auto& lv = c.net->lv[level];
    p.a = c.act(level); p.a_ns = lv.plane * lv.t; p.a_blk = k16Blk; p.a_h = lv.h; p.a_w = lv.w; p.ac0 = ac0; p.cin = cin;
    if (b) { p.saved = c.saved(*b); p.gamma = c.params + b->g; p.beta = c.params + b->b; }
    p.rot = cv.rot; p.rot_n = cv.rot_n;
    p.group_n = c.net->groups > 1 ? c.net->gn : 0; p.gs_saved = c.net->gs_saved;
    p.partial = c.partial(); p.partial_cap = c.net->ws_partial_floats; p.gscale = c.gscale();
}

// the data gradient of a BN -> ReLU -> conv layer: a convolution over the gradient of its outputs with the kEpiDgradBn epilogue
void fill_dgrad(const Ctx16& c, Conv16Params& p, int g_level, int gc0, int level, const Bn16& b, const Conv16& cv) {
    const auto& lg = c.net->lv[g_level];
    const auto& lv = c.net->lv[level];
    p.n = c.net->n; p.h = lv.h; p.w = lv.w;
    p.in = c.dbuf(g_level); p.in_t = lg.t; p.in_blk = k16Blk; p.in_h = lg.h; p.in_w = lg.w; p.in_ns = lg.plane * lg.t; p.ic0 = gc0; p.cin = cv.cout;
    p.wgt = c.w16d(cv);
    p.out = c.dbuf(level); p.out_t = lv.t; p.out_blk = k16Blk; p.out_ns = lv.plane * lv.t; p.oc0 = 0; p.cout = cv.cin;
    p.x = c.act(level); p.x_saved = c.saved(b); p.gamma = c.params + b.g; p.beta = c.params + b.b; p.rot = cv.rot; p.rot_n = cv.rot_n;
    p.out_sums = c.bnsums(b); p.out_sums_slot_stride = c.net->bn_slot_stride;
    p.group_n = c.net->groups > 1 ? c.net->gn : 0; p.gs_saved = c.net->gs_saved; p.gs_out_sums = c.net->gs_saved;
    p.sr_salt = static_cast<unsigned>(cv.w) * 2654435761u;          // a different rounding sequence per layer
}

// dense layer j of a block with `c0` base channels (reads [0, c0 + 12 j), wrote [c0 + 12 j, + 12)): bias gradient + deferred terms,
// weight gradient, and the data gradient with respect to the NEW maps it reads ([c0, c0 + 12 j): they carry the layer-to-layer
// dependency); the base channels of all four layers follow in one pass (dense_block_bwd16)
int dense_wgrad16(const Ctx16& c, int level, int c0, int j, const Bn16& b, const Conv16& cv, hipStream_t side) {
    const auto& lv = c.net->lv[level];
    const int oc0 = c0 + k16Growth * j;
    const double px = static_cast<double>(c.net->n) * lv.plane;
    Wgrad16Params p{};
    p.n = c.net->n; p.h = lv.h; p.w = lv.w;
    fill_wgrad_a(c, p, level, 0, cv.cin, &b, cv);
    p.g = c.dbuf(level); p.g_ns = lv.plane * lv.t; p.g_blk = k16Blk; p.gc0 = oc0; p.cout = cv.cout;
    ProfScope prof(kProfWgradDense, side, 2.0 * px * cv.cin * cv.cout * 9, 2.0 * px * (cv.cin + cv.cout));
    return launch_bf16_wgrad<3>(p, c.grads + cv.w, side);
}

int dense_bwd16(const Ctx16& c, int level, int c0, int j, const Bn16& b, const Conv16& cv, bool with_wgrad = true) {
    const auto& lv = c.net->lv[level];
    const int oc0 = c0 + k16Growth * j;
    int rc = prep_dy16(c, level, oc0, cv.cout, c.grads + cv.b);
    if (rc) return rc;
    const double px = static_cast<double>(c.net->n) * lv.plane;
    if (with_wgrad) {
        hipStream_t side;
        rc = c.fork_wgrad(side);
        if (rc) return rc;
        rc = dense_wgrad16(c, level, c0, j, b, cv, side);
        if (rc) return rc;
    }
    if (j == 0) return 0;
    Conv16Params p{};
    fill_dgrad(c, p, level, oc0, level, b, cv);
    p.oc0 = c0; p.cout = k16Growth * j; p.co_off = c0; p.grp0 = c0 / 48; p.wgroups = (cv.cin + 47) / 48;
    {
        ProfScope prof(kProfDgradDense, c.stream, 2.0 * px * p.cout * cv.cout * 9, px * (6.0 * p.cout + 2.0 * cv.cout));
        rc = launch_bf16_conv<3, 3, kEpiDgradBn, 4, 1, 0, 0, 8>(p, c.stream);          // 4-wave blocks over 8-row tiles: one fits a CU beside a weight-gradient block
    }
    if (rc) return rc;
    return bn_finalize16(c, b, cv, level, c0, k16Growth * j);
}

// a dense block with c0 base channels [0, c0) and its four layers' maps at [c0, c0 + 48)
int dense_block_bwd16(const Ctx16& c, int level, int c0, const Bn16* bn, const Conv16* cv) {
    const bool defer = c.net->wstream && c.net->use_wstream == 2;          // one fork per block: nothing rewrites a prepared G before the join
    for (int j = k16Layers - 1; j >= 0; --j) {
        const int rc = dense_bwd16(c, level, c0, j, bn[j], cv[j], !defer);
        if (rc) return rc;
    }
    if (defer) {
        hipStream_t side;
        int rc = c.fork_wgrad(side);
        if (rc) return rc;
        for (int j = k16Layers - 1; j >= 0; --j) {
            rc = dense_wgrad16(c, level, c0, j, bn[j], cv[j], side);
            if (rc) return rc;
        }
    }
    const auto& lv = c.net->lv[level];
    DgradBlock16Params p{};
    p.n = c.net->n; p.h = lv.h; p.w = lv.w;
    p.g = c.dbuf(level); p.out = c.dbuf(level); p.x = c.act(level); p.ns = lv.plane * lv.t; p.blk = k16Blk; p.gc0 = c0; p.c0 = c0;
    for (int j = 0; j < k16Layers; ++j) {
        p.wgt[j] = c.w16d(cv[j]); p.saved[j] = c.saved(bn[j]); p.gamma[j] = c.params + bn[j].g; p.beta[j] = c.params + bn[j].b;
        p.sums[j] = c.bnsums(bn[j]);
    }
    p.rot = cv[0].rot; p.rot_n = cv[0].rot_n;
    p.group_n = c.net->groups > 1 ? c.net->gn : 0; p.gs_saved = c.net->gs_saved; p.gs_sums = c.net->gs_saved; p.sums_slot_stride = c.net->bn_slot_stride;
    p.sr_salt = static_cast<unsigned>(cv[0].w) * 2246822519u;
    {
        // per base channel and pixel: forward value 2 B, gradient read + written 4 B; the 48 gradient maps 2 B each
        const double px = static_cast<double>(c.net->n) * lv.plane;
        ProfScope prof(kProfDgradDense, c.stream, 2.0 * px * c0 * k16New * 9, px * (6.0 * c0 + 2.0 * k16New));
        const int rc = launch_bf16_dgrad_block(p, c.stream);
        if (rc) return rc;
    }
    static_assert(k16Layers == 4, "bf16_bn_finalize4_kernel");
    BnFin16x4 fin{};
    for (int j = 0; j < k16Layers; ++j) {
        fin.sums[j] = c.bnsums(bn[j]); fin.saved[j] = c.saved(bn[j]); fin.gamma[j] = c.params + bn[j].g;
        fin.ggamma[j] = c.grads + bn[j].g; fin.gbeta[j] = c.grads + bn[j].b; fin.rot[j] = cv[j].rot; fin.rot_n[j] = cv[j].rot_n;
    }
    bf16_bn_finalize4_kernel<<<dim3((c0 + 127) / 128, c.net->groups), 128, 0, c.stream>>>(
        fin, c.pq_p(level), c.pq_q(level), c.gsum(level), c0, static_cast<double>(c.net->gn) * lv.plane, c.training, c.net->gs_saved, c.net->gs_saved,
        2 * lv.t, c.gscale(), c.net->bn_slot_stride);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// transition down (level -> level + 1): reads [0, S), its pooled outputs are channels [0, S) of the next level
int td_bwd16(const Ctx16& c, int level, const Bn16& b, const Conv16& cv) {
    const int next = level + 1;
    const auto& lv = c.net->lv[level];
    const auto& nx = c.net->lv[next];
    int rc = prep_dy16(c, next, 0, cv.cout, c.grads + cv.b);
    if (rc) return rc;
    const uint8_t* idx = reinterpret_cast<const uint8_t*>(c.tape + c.net->idx_off[level]);
    {
        Wgrad16Params p{};
        p.n = c.net->n; p.h = lv.h; p.w = lv.w;
        fill_wgrad_a(c, p, level, 0, cv.cin, &b, cv);
        p.g = c.dbuf(next); p.g_ns = nx.plane * nx.t; p.g_blk = k16Blk; p.gc0 = 0; p.cout = cv.cout; p.g_idx = idx;
        hipStream_t side;
        rc = c.fork_wgrad(side);
        if (rc) return rc;
        rc = launch_bf16_wgrad<1>(p, c.grads + cv.w, side);
        if (rc) return rc;
    }
    Conv16Params p{};
    fill_dgrad(c, p, next, 0, level, b, cv);
    p.ups = 1; p.in_idx = idx;
    rc = launch_bf16_conv<1, 3, kEpiDgradBn, 8, 2, 0, 1>(p, c.stream);
    if (rc) return rc;
    return bn_finalize16(c, b, cv, level);
}

// transition up (src level, channels [src_c0, +48) -> level, channels [S, S + 48)): no BatchNorm, no ReLU
int tu_bwd16(const Ctx16& c, int level, int src_level, int src_c0, const Conv16& cv) {
    const auto& lv = c.net->lv[level];
    const auto& sv = c.net->lv[src_level];
    const int oc0 = c16_skip(level);
    int rc = prep_dy16(c, level, oc0, cv.cout, c.grads + cv.b);
    if (rc) return rc;
    {
        Wgrad16Params p{};
        p.n = c.net->n; p.h = lv.h; p.w = lv.w;
        fill_wgrad_a(c, p, src_level, src_c0, cv.cin, nullptr, cv);
        p.ups = 1;
        p.g = c.dbuf(level); p.g_ns = lv.plane * lv.t; p.g_blk = k16Blk; p.gc0 = oc0; p.cout = cv.cout;
        hipStream_t side;
        rc = c.fork_wgrad(side);
        if (rc) return rc;
        rc = launch_bf16_wgrad<3>(p, c.grads + cv.w, side);
        if (rc) return rc;
    }
    Conv16Params p{};
    p.n = c.net->n; p.h = lv.h; p.w = lv.w;
    p.in = c.dbuf(level); p.in_t = lv.t; p.in_blk = k16Blk; p.in_h = lv.h; p.in_w = lv.w; p.in_ns = lv.plane * lv.t; p.ic0 = oc0; p.cin = cv.cout;
    p.wgt = c.w16d(cv);
    p.out = c.dbuf(src_level); p.out_t = sv.t; p.out_blk = k16Blk; p.out_ns = sv.plane * sv.t; p.oc0 = src_c0; p.cout = cv.cin;
    p.out_sums = c.gsum(src_level) + 2 * src_c0;
    p.sr_salt = static_cast<unsigned>(cv.w) * 3266489917u;
    return launch_bf16_conv<3, 3, kEpiSumPool, 8, 2>(p, c.stream);
}

}  // namespace

// Backward of endo_net16_fwd.  tape: the forward pass's tape, untouched since; grad_out: fp32 [n][1][H][W]; grads: the flat fp32
// parameter-gradient buffer of the fp32 family (same offsets), ACCUMULATED into; ws: endo_net16_bwd_workspace_bytes() bytes, 256-byte
// aligned.  `training` as in the forward call (0: BatchNorm as a fixed affine map, the reference's .eval() backward).  Gradients
// between layers are stored as bf16 (fp32 accumulation inside every kernel); BatchNorm sums, parameter gradients and the deferred
// BatchNorm terms are fp32 / fp64.
extern "C" int N16(bwd)(endo_net16* net, const float* params, const void* tape_, const float* grad_out, float* grads, void* ws_, int training,
                              void* stream_) {
    if (!net || !params || !tape_ || !grad_out || !grads || !ws_) return ENDO_E_BADARG;
    const Table16& tb = table16();
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Ctx16 c{net, params, nullptr, const_cast<char*>(static_cast<const char*>(tape_)), training, stream};
    c.grads = grads; c.ws = static_cast<char*>(ws_);
    if (!net->wstream && net->use_wstream) {
        ENDO_CHECK(hipStreamCreateWithFlags(&net->wstream, hipStreamNonBlocking));
        ENDO_CHECK(hipEventCreateWithFlags(&net->ev_fork, hipEventDisableTiming));
        ENDO_CHECK(hipEventCreateWithFlags(&net->ev_join, hipEventDisableTiming));
    }
    ENDO_CHECK(hipMemsetAsync(c.ws + net->ws_zero_begin, 0, static_cast<size_t>(net->ws_zero_end - net->ws_zero_begin), stream));
    {   // data-gradient weights of the 54 convolutions that have one
        W16Table t{};
        t.layers = static_cast<int>(tb.dconvs.size());
        t.dgrad = 1;
        if (t.layers > 63) return ENDO_E_BADARG;
        int64_t start = 0;
        for (int l = 0; l < t.layers; ++l) {
            const Conv16& cv = *tb.dconvs[l];
            const int64_t groups = (cv.cin + 47) / 48, chunks = (cv.cout + kBfKC - 1) / kBfKC;
            t.start[l] = start; t.w[l] = cv.w; t.out[l] = cv.w16d; t.cout[l] = cv.cin; t.cin[l] = cv.cout; t.cin_k[l] = cv.cout; t.ks[l] = cv.ks; t.nt[l] = 3;
            t.rot[l] = cv.rot; t.rot_n[l] = cv.rot_n; t.base[l] = cv.base; t.koff[l] = cv.koff;
            start += chunks * groups * cv.ks * cv.ks * 3 * 16 * 32;
        }
        t.start[t.layers] = start;
        bf16_all_weights_kernel<<<1024, 256, 0, stream>>>(t, params, reinterpret_cast<uint16_t*>(c.ws + net->ws_w16d));
        ENDO_LAUNCH_CHECK();
    }
    // everything that may fork the side stream runs inside `body`: the join below is reached on every exit path, so an error return never
    // leaves the side stream un-joined with the caller's
    auto body = [&]() -> int {
    int rc;
#ifdef ENDO16_HALF
    ENDO_CHECK(hipMemsetAsync(c.ws + net->ws_gscale, 0, 16, stream));
    {
        const int64_t count = static_cast<int64_t>(net->n) * net->lv[0].plane;
        int gb = static_cast<int>((count + 8191) / 8192);
        gb = gb > 256 ? 256 : gb;
        s16_grad_max_kernel<<<gb, 1024, 0, stream>>>(grad_out, count, reinterpret_cast<float*>(c.ws + net->ws_gscale));
        s16_grad_scale_kernel<<<1, 64, 0, stream>>>(reinterpret_cast<float*>(c.ws + net->ws_gscale));
    }
    ENDO_LAUNCH_CHECK();
#endif
    {
        const auto& lv = net->lv[0];
        int bx = static_cast<int>((lv.plane * 4 + 255) / 256);          // 4 lanes per pixel
        bx = bx > 512 ? 512 : bx;
        bf16_final_bwd_kernel<<<dim3(bx, net->n), 256, 0, stream>>>(grad_out, reinterpret_cast<const float*>(c.tape + net->pre_off), c.act(0), c.dbuf(0),
                                                                    lv.plane * lv.t, static_cast<int>(lv.plane), params + tb.final_.w, tb.final_.rot,
                                                                    tb.final_.rot_n, grads + tb.final_.w, grads + tb.final_.b, c.gsum(0), c.gscale());
        ENDO_LAUNCH_CHECK();
    }
    for (int i = k16Levels - 1; i >= 0; --i) {
        const int l = k16Levels - 1 - i, src = l + 1;
        rc = dense_block_bwd16(c, l, c16_skip(l) + k16New, tb.up_bn[i], tb.up_conv[i]);
        if (rc) return rc;
        rc = tu_bwd16(c, l, src, (i == 0) ? 288 : c16_skip(src) + k16New, tb.tu_conv[i]);
        if (rc) return rc;
    }
    rc = dense_block_bwd16(c, k16Levels, 288, tb.bott_bn, tb.bott_conv);
    if (rc) return rc;
    for (int l = k16Levels - 1; l >= 0; --l) {
        rc = td_bwd16(c, l, tb.td_bn[l], tb.td_conv[l]);
        if (rc) return rc;
        rc = dense_block_bwd16(c, l, c16_down_in(l), tb.down_bn[l], tb.down_conv[l]);
        if (rc) return rc;
    }
    {   // first convolution: bias and weight gradient (the image needs none)
        rc = prep_dy16(c, 0, 0, k16First, grads + tb.first.b);
        if (rc) return rc;
        const auto& lv = net->lv[0];
        Wgrad16Params p{};
        p.n = net->n; p.h = lv.h; p.w = lv.w;
        p.a = reinterpret_cast<const uint16_t*>(c.tape + net->in_off); p.a_ns = lv.plane * 8; p.a_blk = 8; p.a_h = lv.h; p.a_w = lv.w; p.ac0 = 0; p.cin = 4;
        p.cin_w = 3;
        p.g = c.dbuf(0); p.g_ns = lv.plane * lv.t; p.g_blk = k16Blk; p.gc0 = 0; p.cout = k16First;
        p.partial = c.partial(); p.partial_cap = c.net->ws_partial_floats; p.gscale = c.gscale();
        hipStream_t side;
        rc = c.fork_wgrad(side);
        if (rc) return rc;
        rc = launch_bf16_wgrad<3>(p, grads + tb.first.w, side);
        if (rc) return rc;
    }
    return 0;
    };
    const int rc_body = body();
    if (net->wstream && net->use_wstream) {          // join: the caller's stream continues only after every weight gradient has landed
        ENDO_CHECK(hipEventRecord(net->ev_join, net->wstream));
        ENDO_CHECK(hipStreamWaitEvent(stream, net->ev_join, 0));
    }
    return rc_body;
}
