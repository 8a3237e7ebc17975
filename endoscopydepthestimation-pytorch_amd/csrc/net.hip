// FC-DenseNet57 forward / backward schedule on the HIP kernels of conv_kernels.h / wgrad_kernels.h.
//
// Data layout in HBM (all fp32, NCHW planar, caller-owned):
//   one "level buffer" U_L per resolution level L = 0..4 (H>>L x W>>L) holding, as channel planes,
//       [0,48)              transition-up output of the up path
//       [48, 48+C_L)        down-block input            (C_L = 48 + 48 L)
//       [48+C_L, 96+C_L)    the 4 x 12 maps the down block adds
//       [96+C_L, 144+C_L)   the 4 x 12 maps the up block adds
//   and U_5 (bottleneck, H>>5) = [0,288) input, [288,336) new maps.
//   Every convolution reads a channel range of a level buffer and writes a disjoint range of the
//   same (or the next) level buffer, so torch.cat (reference models.py:46,48,52,79) never happens:
//   the skip connection is simply the range [48, 96+C_L) that both paths address.
//   The gradient workspace mirrors this layout.
// Training-mode BatchNorm: per-channel sum / sum^2 (fp64) are produced once, by the epilogue of the
// kernel that writes the channel, and consumed by every later BN over it (each with its own
// gamma/beta and running statistics).
#include <new>
#include <vector>

#include "dgrad_kernels.h"
#include "dgrad_block_kernels.h"
#include "dgrad_newmap_kernels.h"
#include "td_dgrad_kernels.h"
#include "td_fwd_kernels.h"
#include "wgrad_taps_kernels.h"
#include "wgrad1x1_kernels.h"
#include "wgrad_nsplit_kernels.h"
#include "wgrad_x3_kernels.h"
#include "wgrad_f34_kernels.h"
#include "wgrad_subpix_kernels.h"
#include "wino_fwd_kernels.h"
#include "wino4_fwd_kernels.h"
#include "dgrad_wino_kernels.h"
#include "dgrad_wino3_kernels.h"

#include <cstdlib>

namespace endo {

int run_dgrad_wino3_nl4(const DgradBlockParams& p, const float* const* u, hipStream_t stream);          // dgrad_wino3.hip
bool dgrad_wino3p_applies(const DgradBlockParams& p);
int run_dgrad_wino3p_nl4(const DgradBlockParams& p, const float* const* u, int blocks, double* fw_parts, int* blocks_used, hipStream_t stream);

constexpr int kGrowth = 12;
constexpr int kLayers = 4;
constexpr int kFirst = 48;
constexpr int kLevels = 5;
constexpr int kNew = kGrowth * kLayers;   // 48
constexpr float kBnEps = 1.0e-5f;
constexpr float kBnMomentum = 0.1f;
constexpr int kSideStreamFromLevel = 0;      // weight gradients of levels >= this run on the side stream (in-job A/B: 0 -> -4.7 %, 1 -> -3 %, 2 -> -1.8 % step time)

struct ConvP { int64_t w, b; int cout, cin, ks; int64_t u; int64_t ud; };      // u / ud: offsets of the layer's Winograd-domain forward / data-gradient weights (dense layers), floats
struct BnP { int64_t g, b; int c; int64_t run; int64_t saved; };   // run: offset in bn_running; saved: offset (pairs) in saved/scratch

struct Table {
    ConvP first, final_;
    ConvP down_conv[kLevels][kLayers], td_conv[kLevels], bott_conv[kLayers], tu_conv[kLevels], up_conv[kLevels][kLayers];
    BnP down_bn[kLevels][kLayers], td_bn[kLevels], bott_bn[kLayers], up_bn[kLevels][kLayers];
    std::vector<int64_t> param_offsets;   // 210 tensors in .parameters() order
    std::vector<BnP*> bn_order;           // 49 BN layers in module order
    int64_t param_floats = 0, bn_floats = 0, bn_width_total = 0;
    WinoWeightTable wino;                 // the 44 dense layers' forward weights in Winograd form (wino_fwd_kernels.h)
    WinoWeightTable wino4;                // the same layers in F(4x4, 3x3) form (wino4_fwd_kernels.h): u_off in floats of ITS buffer
    int64_t wino4_floats = 0;
    int64_t wino_floats = 0;
    WinoDgradTable wino_dgrad;            // and their data-gradient weights over the base channels of their block (dgrad_wino_kernels.h)
    int64_t wino_dgrad_floats = 0;
};

inline int down_in(int level) { return kFirst + kNew * level; }          // C_L
inline int level_channels(int level) { return level < kLevels ? 144 + down_in(level) : 288 + kNew; }

static const Table& table() {
    static Table* t = [] {
        Table* tb = new Table();
        int64_t off = 0;
        auto conv = [&](ConvP& c, int cout, int cin, int ks) {
            c.cout = cout; c.cin = cin; c.ks = ks; c.u = -1; c.ud = -1;
            c.w = off; tb->param_offsets.push_back(off); off += static_cast<int64_t>(cout) * cin * ks * ks;
            c.b = off; tb->param_offsets.push_back(off); off += cout;
        };
        int64_t run = 0, saved = 0;
        auto bn = [&](BnP& b, int c) {
            b.c = c;
            b.g = off; tb->param_offsets.push_back(off); off += c;
            b.b = off; tb->param_offsets.push_back(off); off += c;
            b.run = run; run += 2 * c;
            b.saved = saved; saved += c;
            tb->bn_order.push_back(&b);
        };
        conv(tb->first, kFirst, 3, 3);
        for (int l = 0; l < kLevels; ++l)
            for (int j = 0; j < kLayers; ++j) { bn(tb->down_bn[l][j], down_in(l) + kGrowth * j); conv(tb->down_conv[l][j], kGrowth, down_in(l) + kGrowth * j, 3); }
        for (int l = 0; l < kLevels; ++l) { bn(tb->td_bn[l], down_in(l) + kNew); conv(tb->td_conv[l], down_in(l) + kNew, down_in(l) + kNew, 1); }
        for (int j = 0; j < kLayers; ++j) { bn(tb->bott_bn[j], 288 + kGrowth * j); conv(tb->bott_conv[j], kGrowth, 288 + kGrowth * j, 3); }
        for (int i = 0; i < kLevels; ++i) conv(tb->tu_conv[i], kNew, kNew, 3);
        for (int i = 0; i < kLevels; ++i) {
            const int l = kLevels - 1 - i;
            for (int j = 0; j < kLayers; ++j) { const int cin = 96 + down_in(l) + kGrowth * j; bn(tb->up_bn[i][j], cin); conv(tb->up_conv[i][j], kGrowth, cin, 3); }
        }
        conv(tb->final_, 1, 192, 1);
        tb->param_floats = off;
        tb->bn_floats = run;
        tb->bn_width_total = saved;
        {   // Winograd-domain weights of the dense layers: kWinoUStride floats per input channel, layer after layer
            WinoWeightTable& wt = tb->wino;
            wt.layers = 0; wt.start[0] = 0;
            int64_t uoff = 0;
            auto add = [&](ConvP& c) {
                const int l = wt.layers++;
                c.u = uoff;
                wt.cin[l] = c.cin; wt.cout[l] = c.cout; wt.w_off[l] = c.w; wt.u_off[l] = uoff;
                wt.start[l + 1] = wt.start[l] + c.cin * 16;
                uoff += static_cast<int64_t>(c.cin) * kWinoUStride;
            };
            for (int l = 0; l < kLevels; ++l) for (int j = 0; j < kLayers; ++j) add(tb->down_conv[l][j]);
            for (int j = 0; j < kLayers; ++j) add(tb->bott_conv[j]);
            for (int i = 0; i < kLevels; ++i) for (int j = 0; j < kLayers; ++j) add(tb->up_conv[i][j]);
            tb->wino_floats = uoff;
            tb->wino4 = wt;
            int64_t u4 = 0;
            for (int l = 0; l < wt.layers; ++l) { tb->wino4.u_off[l] = u4; u4 += static_cast<int64_t>(wt.cin[l]) * kW4UStride; }
            tb->wino4_floats = u4;
        }
        {   // data-gradient weights in Winograd form: per dense block, the 16-channel groups of its base channels, for each of its layers
            WinoDgradTable& wd = tb->wino_dgrad;
            wd.layers = 0; wd.start[0] = 0;
            int64_t uoff = 0;
            auto add = [&](ConvP& c, int base) {
                const int l = wd.layers++;
                c.ud = uoff;
                wd.cin[l] = c.cin; wd.groups[l] = base / 16; wd.w_off[l] = c.w; wd.u_off[l] = uoff;
                wd.start[l + 1] = wd.start[l] + 16 * wd.groups[l] * 12;
                uoff += static_cast<int64_t>(wd.groups[l]) * kWinoDgradSlice;
            };
            for (int l = 0; l < kLevels; ++l) for (int j = 0; j < kLayers; ++j) add(tb->down_conv[l][j], down_in(l));
            for (int j = 0; j < kLayers; ++j) add(tb->bott_conv[j], 288);
            for (int i = 0; i < kLevels; ++i) for (int j = 0; j < kLayers; ++j) add(tb->up_conv[i][j], 96 + down_in(kLevels - 1 - i));
            tb->wino_dgrad_floats = uoff;
        }
        return tb;
    }();
    return *t;
}

}  // namespace endo

using namespace endo;

struct endo_net {
    int n, h, w;           // n = samples per group
    int groups;            // independent forward / backward passes batched into every launch (each with its own BN statistics)
    int64_t gs;            // floats between two groups' tapes and between their gradient workspaces
    struct Level { int h, w, t; int64_t plane; int64_t act, grad; int64_t sums; int64_t pq; } lv[kLevels + 1];
    int64_t pre_off;       // final conv pre-activation (floats, tape)
    int64_t saved_off;     // BN saved mean/rstd (floats, tape), 2 per BN channel
    int64_t idx_off[kLevels];   // pool argmax codes (byte offsets from tape base)
    int64_t sums_off;      // fp64 per-channel sums (byte offset, 8-aligned)
    int64_t sums_bytes;
    int64_t tape_floats;
    int64_t partial_off;   // split-K partial sums of the coarse-level dense layers (floats, tape)
    int64_t wino_off;      // Winograd-domain forward weights of the dense layers (floats, tape; group 0's copy serves all groups)
    int64_t wino4_off;     // the same in F(4x4, 3x3) form
    int64_t pq_off;        // floats, gradws: P then Q per level channel
    int64_t pq_floats;
    int64_t scratch_off;   // byte offset in gradws of fp64 BN scratch
    int64_t scratch_bytes;
    int64_t slot_stride;   // doubles between the kBnSlots copies of the BN scratch (common.h)
    int64_t wg_scratch_off;   // float offset in gradws of the weight-gradient partial sums (wgrad_nsplit_kernels.h)
    int64_t tuw_scratch_off;  // float offset in gradws of the transition-up data-gradient weights (tu_subpix_dgrad_weights_kernel)
    int64_t wd_off;           // float offset in gradws of the Winograd-domain data-gradient weights (group 0's copy serves all groups)
    int64_t gplane_off;       // float offset in gradws of g = grad_out * sign(pre), one plane per sample (final_g_kernel)
    int64_t bias_parts_off;   // float offset in gradws (group 0's) of prep_dy's per-block sums of G (BiasParts), bias_parts_floats long
    int64_t fw_parts_off;     // float offset in gradws (group 0's) of the persistent base pass's final-conv weight partials (kFwPartBlocks x 192 doubles)
    int cus;                  // compute units of the device the handle was created on: the persistent kernels launch one block per CU
    int64_t bias_parts_floats;
    int64_t gradws_floats;
    // Weight gradients run on a side stream: a layer's wgrad depends only on its prepared dY and the forward tape, nothing on the
    // backward chain depends on it (it only adds into the flat gradient), so it overlaps the data-gradient chain -- which at the
    // coarse levels is a string of launches too small to fill the chip.  Forked after every prep_dy, joined once at the end.
    hipStream_t wstream;
    hipEvent_t ev_fork, ev_join;
    // kernel-form / precision options of THIS network (endo_net_set_option): two networks in one process never change each
    // other's arithmetic, and a thread stepping one model is not affected by another thread configuring a second one
    int opt[ENDO_OPT_COUNT];
};

namespace endo {

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------

// Conv-bias gradients (sum of the prepared G over pixels and samples, train.py's loss.backward() through models.py's convolutions): prep_dy's blocks
// used to add their sums to bias_grad[c] with one float atomic each -- 3 840 atomics of a level-0 launch on the ONE cache line that holds a
// layer's 12 bias gradients, served one after the other at ~8 ns: 31 of the launch's 57 us (tools/prep_bench: 24.7 us without them, the rate of a
// plain read-2-write-1 pass).  Now a block stores its sum (plain store, its own slot) and ONE launch at the end of the backward pass adds up
// the slots of every prep_dy launch in a fixed order (bias_reduce_kernel): no serialised atomics, and the same bits in every run.
constexpr int kBiasPartChannels = 2048;          // channels prepared per backward pass: 1 776 at FC-DenseNet57 (all conv outputs but the final one)
constexpr int kBiasPartLaunches = 64;
constexpr int kFwPartBlocks = 512;          // persistent blocks whose final-conv weight partials fit the workspace (dgrad_wino3p_kernels.h, FW)
struct BiasReduceTable {
    int n;
    struct Entry { const float* parts; float* bias; int count; int nparts; } e[kBiasPartLaunches];
};
struct BiasParts {          // owned by endo_net_bwd's frame, filled by prep_dy
    BiasReduceTable table;
    int64_t used;           // floats of the region handed out
};

// per BN layer (up to four): its backward sums, saved statistics, parameters and parameter gradients, offset to the first
// channel of the range a launch works on
struct BnFin4 {
    const double* scratch[4];
    const float* saved[4];
    const float* gamma[4];
    float* ggamma[4];
    float* gbeta[4];
};

// G = dbuf + P x + Q in place for a channel range (the deferred mean terms of every BN that consumed
// the channel), and bias gradient += sum G.
// nl > 0: the range's consumers INSIDE its dense block (fin: nl BN layers) have just been differentiated by one fused pass and
// their deferred terms are not in P, Q yet: every block derives them from the pass's sums (the arithmetic of
// bn_bwd_finalize4_kernel, term by term), and the first block of a (channel, group) also adds the BN parameter gradients.
// This takes the separate finalize launch between the pass and this kernel off the backward chain.
// vg: the range's raw gradient is not in dbuf but is the final convolution's rank-one data gradient g(pixel) * vw[channel]
// (DgradBlockParams::vg): read the one plane g instead (first writer of the range).
__global__ void __launch_bounds__(256) prep_dy_kernel(float* __restrict__ dbuf, const float* __restrict__ x, int64_t ns, int plane,
                                                      const float* __restrict__ pq_p, const float* __restrict__ pq_q,
                                                      float* bias_grad, int group_n, int64_t gs, const BnFin4 fin, int nl,
                                                      double count, int training, int64_t slot_stride,
                                                      const float* __restrict__ vg, const float* __restrict__ vw, float* __restrict__ parts) {
    __shared__ double scratch[4];
    const int c = blockIdx.y;
    const int grp = blockIdx.z / group_n, n = blockIdx.z - grp * group_n;      // grouped batch: per-group buffers, shared bias gradient
    float pc = pq_p[grp * gs + c], qc = pq_q[grp * gs + c];
    if (nl > 0) {
        const int64_t go = grp * gs;
        double dp = 0.0, dq = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= nl) break;
            const double s1 = bn_slot_sum(fin.scratch[j] + go / 2 + 2 * c, slot_stride);          // block-uniform addresses
            const double s2 = bn_slot_sum(fin.scratch[j] + go / 2 + 2 * c + 1, slot_stride);
            if (blockIdx.x == 0 && n == 0 && threadIdx.x == 0) {
                atomicAdd(fin.ggamma[j] + c, static_cast<float>(s2));
                atomicAdd(fin.gbeta[j] + c, static_cast<float>(s1));
            }
            if (training) {
                const double mean = fin.saved[j][go + 2 * c], rstd = fin.saved[j][go + 2 * c + 1];
                const double scale = fin.gamma[j][c] * rstd;
                const double k = scale * rstd * s2 / count;
                dp += static_cast<double>(static_cast<float>(-k));
                dq += static_cast<double>(static_cast<float>(-scale * s1 / count + k * mean));
            }
        }
        pc += static_cast<float>(dp);
        qc += static_cast<float>(dq);
    }
    const int64_t base = grp * gs + n * ns + static_cast<int64_t>(c) * plane;
    const float* vg_n = vg ? vg + grp * gs + static_cast<int64_t>(n) * plane : nullptr;          // (block-uniform)
    const float wf = vg ? vw[c] : 0.f;
    float part = 0.f;
    if ((plane & 3) == 0 && !vg_n) {
        // two iterations' loads in flight per thread (round 5: +0.2 % on the step in three alternating in-job runs; non-temporal loads of x: +-0)
        const int stride = gridDim.x * blockDim.x * 4;
        int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
        for (; i + stride < plane; i += 2 * stride) {
            f32x4 g0 = *reinterpret_cast<const f32x4*>(dbuf + base + i);
            f32x4 g1 = *reinterpret_cast<const f32x4*>(dbuf + base + i + stride);
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + base + i);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(x + base + i + stride);
#pragma unroll
            for (int e = 0; e < 4; ++e) { g0[e] += fmaf(pc, x0[e], qc); part += g0[e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) { g1[e] += fmaf(pc, x1[e], qc); part += g1[e]; }
            *reinterpret_cast<f32x4*>(dbuf + base + i) = g0;
            *reinterpret_cast<f32x4*>(dbuf + base + i + stride) = g1;
        }
        for (; i < plane; i += stride) {
            f32x4 g = *reinterpret_cast<const f32x4*>(dbuf + base + i);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) { g[e] += fmaf(pc, xv[e], qc); part += g[e]; }
            *reinterpret_cast<f32x4*>(dbuf + base + i) = g;
        }
    } else if ((plane & 3) == 0) {
        for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4; i < plane; i += gridDim.x * blockDim.x * 4) {
            f32x4 g;
            if (vg_n) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(vg_n + i);
                g = f32x4{gv[0] * wf, gv[1] * wf, gv[2] * wf, gv[3] * wf};
            } else {
                g = *reinterpret_cast<const f32x4*>(dbuf + base + i);
            }
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) { g[e] += fmaf(pc, xv[e], qc); part += g[e]; }
            *reinterpret_cast<f32x4*>(dbuf + base + i) = g;
        }
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
            const float g = (vg_n ? vg_n[i] * wf : dbuf[base + i]) + fmaf(pc, x[base + i], qc);
            dbuf[base + i] = g;
            part += g;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double v = wave_sum(static_cast<double>(part));
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0 && bias_grad) {
        const float t = static_cast<float>(scratch[0] + scratch[1] + scratch[2] + scratch[3]);
        // parts: [channel][sample of all groups][block]: added up by bias_reduce_kernel at the end of the backward pass (BiasParts)
        if (parts) parts[(static_cast<int64_t>(c) * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x] = t;
        else atomicAdd(bias_grad + c, t);
    }
}

// bias[c] += sum of prep_dy's per-block sums, one wave per (launch, channel), fixed order
__global__ void __launch_bounds__(256) bias_reduce_kernel(const BiasReduceTable tb) {
    const BiasReduceTable::Entry& e = tb.e[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = blockIdx.y * 4 + wave; c < e.count; c += gridDim.y * 4) {
        const float* src = e.parts + static_cast<int64_t>(c) * e.nparts;
        double t = 0.0;
        for (int i = lane; i < e.nparts; i += 64) t += static_cast<double>(src[i]);
        t = wave_sum(t);
        if (lane == 0) e.bias[c] += static_cast<float>(t);
    }
}

// BN parameter gradients from the dgrad epilogue's sums, and the deferred dx terms:
//   dx = scale*dz - scale*S1/M - scale*rstd*(S2/M)*(x - mean)   =>   P += -scale*rstd*S2/M,
//   Q += -scale*S1/M + scale*rstd*S2/M*mean          (S1 = sum dz, S2 = sum dz*xhat)
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ scratch, const float* __restrict__ saved,
                                       const float* __restrict__ gamma, float* __restrict__ ggamma, float* __restrict__ gbeta,
                                       float* __restrict__ pq_p, float* __restrict__ pq_q, int c_count, double count, int training,
                                       int64_t gs, int64_t slot_stride) {
    // blockIdx.y = sample group: its own sums, statistics and deferred terms; the parameter gradients add up over groups
    scratch += blockIdx.y * (gs / 2); saved += blockIdx.y * gs; pq_p += blockIdx.y * gs; pq_q += blockIdx.y * gs;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < c_count; c += gridDim.x * blockDim.x) {
        const double s1 = bn_slot_sum(scratch + 2 * c, slot_stride), s2 = bn_slot_sum(scratch + 2 * c + 1, slot_stride);
        atomicAdd(ggamma + c, static_cast<float>(s2));
        atomicAdd(gbeta + c, static_cast<float>(s1));
        if (training) {
            const double mean = saved[2 * c], rstd = saved[2 * c + 1];
            const double scale = gamma[c] * rstd;
            const double k = scale * rstd * s2 / count;
            pq_p[c] += static_cast<float>(-k);
            pq_q[c] += static_cast<float>(-scale * s1 / count + k * mean);
        }
    }
}

// the same for up to four BN layers of a dense block over channels they share: one launch, one read-modify-write of
// P and Q
__global__ void bn_bwd_finalize4_kernel(const BnFin4 a, int nl, float* __restrict__ pq_p, float* __restrict__ pq_q, int c_count, double count,
                                        int training, int64_t gs, int64_t slot_stride) {
    const int64_t go = blockIdx.y * gs;          // sample group offset (floats)
    pq_p += go; pq_q += go;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < c_count; c += gridDim.x * blockDim.x) {
        double dp = 0.0, dq = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= nl) break;          // nl = layers in use (block-uniform)
            const double s1 = bn_slot_sum(a.scratch[j] + go / 2 + 2 * c, slot_stride);
            const double s2 = bn_slot_sum(a.scratch[j] + go / 2 + 2 * c + 1, slot_stride);
            atomicAdd(a.ggamma[j] + c, static_cast<float>(s2));
            atomicAdd(a.gbeta[j] + c, static_cast<float>(s1));
            if (training) {
                const double mean = a.saved[j][go + 2 * c], rstd = a.saved[j][go + 2 * c + 1];
                const double scale = a.gamma[j][c] * rstd;
                const double k = scale * rstd * s2 / count;
                // same rounding sequence as four single-layer finalizes: each term is rounded to fp32 before it is added
                dp += static_cast<double>(static_cast<float>(-k));
                dq += static_cast<double>(static_cast<float>(-scale * s1 / count + k * mean));
            }
        }
        if (training) {
            pq_p[c] += static_cast<float>(dp);
            pq_q[c] += static_cast<float>(dq);
        }
    }
}

// Tap-summed weights of the transition-up sub-pixel phases (conv_dma_kernels.h, PH) in the compact layout the kernel
// DMAs: out[a][ci][400] with, for row tap t in {0,1} and slot in [0,12): slot 0-2 = column tap 0 of the b = 0 tiles,
// 3-8 = column tap 1 of all six tiles (b = 0 | b = 1), 9-11 = column tap 2 of the b = 1 tiles; 16 channels per tile.
// Row phase 0 sees input rows (y-1, y): row tap 0 <- ky 0, 1 <- ky 1 + ky 2; row phase 1 sees (y, y+1): 0 <- ky 0 + ky 1,
// 1 <- ky 2.  Column phases likewise: b = 0 sees (x-1, x), b = 1 sees (x, x+1).
__global__ void tu_phase_weights_kernel(const float* __restrict__ w, int cout, int cin, float* __restrict__ out) {
    constexpr int kPhW = 2 * 12 * 16 + 16;
    const int total = 2 * cin * 2 * 12 * 16;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane16 = i & 15;
        const int slot = (i >> 4) % 12;
        const int t = (i / (16 * 12)) & 1;
        const int ci = (i / (16 * 12 * 2)) % cin;
        const int a = i / (16 * 12 * 2 * cin);
        int ctap, tile;                                     // column tap (0..2 on the low-res grid) and tile (0..5)
        if (slot < 3) { ctap = 0; tile = slot; } else if (slot < 9) { ctap = 1; tile = slot - 3; } else { ctap = 2; tile = slot - 9 + 3; }
        const int b = tile / 3, co = (tile % 3) * 16 + lane16;
        float v = 0.f;
        if (co < cout) {
            const float* src = w + (static_cast<int64_t>(co) * cin + ci) * 9;
            // original rows / columns that fall on this low-res tap
            const int ky0 = a == 0 ? (t == 0 ? 0 : 1) : (t == 0 ? 0 : 2), ky1 = a == 0 ? (t == 0 ? 0 : 2) : (t == 0 ? 1 : 2);
            int kx0, kx1;
            if (b == 0) { kx0 = ctap == 0 ? 0 : 1; kx1 = ctap == 0 ? 0 : 2; }          // b = 0: tap 0 <- kx 0, tap 1 <- kx 1 + kx 2
            else { kx0 = ctap == 1 ? 0 : 2; kx1 = ctap == 1 ? 1 : 2; }                  // b = 1: tap 1 <- kx 0 + kx 1, tap 2 <- kx 2
            for (int ky = ky0; ky <= ky1; ++ky)
                for (int kx = kx0; kx <= kx1; ++kx) v += src[ky * 3 + kx];
        }
        out[(static_cast<int64_t>(a) * cin + ci) * kPhW + (t * 12 + slot) * 16 + lane16] = v;
    }
}

// Tap-summed weights of the transition-up DATA gradient in sub-pixel form (conv_dma_kernels.h, IN_SUBPIX):
// out[pseudo = (alpha, beta, co)][208] = [(tyi, txi)][ci] (+ pad).  With u = a + 1 - ky the row offset of dY seen from low-res
// row y:  alpha = 0: tyi 0 (row y) <- ky 1 + ky 2, tyi 1 (row y+1) <- ky 0;  alpha = 1: tyi 0 (row y-1) <- ky 2,
// tyi 1 (row y) <- ky 0 + ky 1.  Columns likewise.
__global__ void tu_subpix_dgrad_weights_kernel(const float* __restrict__ w, int cout, int cin, float* __restrict__ out) {
    const int pitch = 4 * cin + 16;
    const int total = 4 * cout * 4 * cin;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i % cin;
        const int tap = (i / cin) & 3;
        const int co = (i / (4 * cin)) % cout;
        const int ph = i / (4 * cin * cout);
        const int alpha = ph >> 1, beta = ph & 1, tyi = tap >> 1, txi = tap & 1;
        const int ky0 = alpha == 0 ? (tyi == 0 ? 1 : 0) : (tyi == 0 ? 2 : 0), ky1 = alpha == 0 ? (tyi == 0 ? 2 : 0) : (tyi == 0 ? 2 : 1);
        const int kx0 = beta == 0 ? (txi == 0 ? 1 : 0) : (txi == 0 ? 2 : 0), kx1 = beta == 0 ? (txi == 0 ? 2 : 0) : (txi == 0 ? 2 : 1);
        const float* src = w + (static_cast<int64_t>(co) * cin + ci) * 9;
        float v = 0.f;
        for (int ky = ky0; ky <= ky1; ++ky)
            for (int kx = kx0; kx <= kx1; ++kx) v += src[ky * 3 + kx];
        out[static_cast<int64_t>(ph * cout + co) * pitch + tap * cin + ci] = v;
    }
}

// split-K epilogue of the coarse-level dense layers: out = bias + sum over K slices of the partial sums,
// plus the per-channel sum / sum^2 that later BN layers need.  grid (x blocks, channel, sample).
__global__ void __launch_bounds__(256) finalize_partial_kernel(const float* __restrict__ partial, int64_t split_stride, int ksplit,
                                                               int64_t pns, int plane, const float* __restrict__ bias,
                                                               float* __restrict__ out, int64_t out_ns, double* out_sums,
                                                               int group_n, int64_t gs) {
    __shared__ double scratch[2 * 4];
    const int c = blockIdx.y;
    const int grp = blockIdx.z / group_n, n = blockIdx.z - grp * group_n;
    partial += grp * gs; out += grp * gs;
    if (out_sums) out_sums += grp * (gs / 2);
    const float b = bias[c];
    float part[2] = {0.f, 0.f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        float v = b;
        for (int s = 0; s < ksplit; ++s) v += partial[s * split_stride + n * pns + static_cast<int64_t>(c) * plane + i];
        out[n * out_ns + static_cast<int64_t>(c) * plane + i] = v;
        part[0] += v;
        part[1] += v * v;
    }
    if (out_sums) block_sum_atomic<2>(part, out_sums + 2 * c, scratch);          // (block-uniform) no statistics in inference mode
}

// final 1x1 conv 192 -> 1 and |.| (reference models.py:186).  HBM-bound: reads each plane once.
// c_first > 0: `pre` already holds the sum over channels [0, c_first) (wino4_fwd_kernel<true>, the last dense layer's launch) and only the rest is read
__global__ void __launch_bounds__(256) final_fwd_kernel(const float* __restrict__ u, int64_t ns, int plane, int cin,
                                                        const float* __restrict__ wgt, const float* __restrict__ bias,
                                                        float* __restrict__ pre, float* __restrict__ out, int group_n, int64_t gs, int c_first) {
    __shared__ float s_w[192];
    for (int c = threadIdx.x; c < cin; c += blockDim.x) s_w[c] = wgt[c];
    __syncthreads();
    // u and pre live in the sample group's tape, out is the caller's [groups * group_n] tensor
    const int grp = blockIdx.y / group_n, n = blockIdx.y - grp * group_n;
    u += grp * gs;
    if (pre) pre += grp * gs;
    out += static_cast<int64_t>(grp) * group_n * plane;
    const bool vec = (plane & 3) == 0;
    if (vec) {
        for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4; i < plane; i += gridDim.x * blockDim.x * 4) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (c_first > 0) acc = *reinterpret_cast<const f32x4*>(pre + static_cast<int64_t>(n) * plane + i);
            const float* src = u + n * ns + i;
            for (int c = c_first; c < cin; ++c) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + static_cast<int64_t>(c) * plane);
                const float wc = s_w[c];
                acc[0] = fmaf(v[0], wc, acc[0]); acc[1] = fmaf(v[1], wc, acc[1]);
                acc[2] = fmaf(v[2], wc, acc[2]); acc[3] = fmaf(v[3], wc, acc[3]);
            }
            const float b = bias[0];
            f32x4 o;
            for (int e = 0; e < 4; ++e) { acc[e] += b; o[e] = fabsf(acc[e]); }
            if (pre) *reinterpret_cast<f32x4*>(pre + static_cast<int64_t>(n) * plane + i) = acc;
            *reinterpret_cast<f32x4*>(out + static_cast<int64_t>(n) * plane + i) = o;
        }
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
            float acc = c_first > 0 ? pre[static_cast<int64_t>(n) * plane + i] : 0.f;
            for (int c = c_first; c < cin; ++c) acc = fmaf(u[n * ns + static_cast<int64_t>(c) * plane + i], s_w[c], acc);
            acc += bias[0];
            if (pre) pre[static_cast<int64_t>(n) * plane + i] = acc;
            out[static_cast<int64_t>(n) * plane + i] = fabsf(acc);
        }
    }
}

__device__ __forceinline__ float sign_of(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// g = gout * sign(pre): the one plane the final convolution's data gradient g * w[c] is made of (DgradBlockParams::vg) -- the kernels of the
// last up block form the 192 products themselves instead of reading them back from 192 planes
__global__ void __launch_bounds__(256) final_g_kernel(const float* __restrict__ gout, const float* __restrict__ pre, float* __restrict__ g,
                                                      int plane, int group_n, int64_t gs) {
    const int grp = blockIdx.y / group_n, n = blockIdx.y - grp * group_n;
    gout += static_cast<int64_t>(blockIdx.y) * plane; pre += grp * gs + static_cast<int64_t>(n) * plane; g += grp * gs + static_cast<int64_t>(n) * plane;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) g[i] = gout[i] * sign_of(pre[i]);
}

// dbuf[c] = (gout * sign(pre)) * w[c] for planes [c_first, cin) (first writer of the level-0 gradient buffer)
__global__ void __launch_bounds__(256) final_bwd_data_kernel(const float* __restrict__ gout, const float* __restrict__ pre,
                                                             const float* __restrict__ wgt, float* __restrict__ dbuf,
                                                             int64_t ns, int plane, int cin, int group_n, int64_t gs, int c_first) {
    __shared__ float s_w[192];
    for (int c = threadIdx.x; c < cin; c += blockDim.x) s_w[c] = wgt[c];
    __syncthreads();
    const int grp = blockIdx.y / group_n, n = blockIdx.y - grp * group_n;
    gout += static_cast<int64_t>(grp) * group_n * plane; pre += grp * gs; dbuf += grp * gs;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += gridDim.x * blockDim.x) {
        const float g = gout[static_cast<int64_t>(n) * plane + i] * sign_of(pre[static_cast<int64_t>(n) * plane + i]);
        for (int c = c_first; c < cin; ++c) dbuf[n * ns + static_cast<int64_t>(c) * plane + i] = g * s_w[c];
    }
}

// dw[c] += sum g * u[c]; channel index cin is the bias (sum g).  grid (cin + 1, slices, n): float4 streaming,
// two independent accumulators; g = gout * sign(pre) is recomputed from two L2-resident planes.
// c_first: the channels below it are left to the last up block's persistent base pass (dgrad_wino3p_kernels.h, FW); grid.x = cin - c_first + 1
__global__ void __launch_bounds__(256) final_bwd_weight_kernel(const float* __restrict__ gout, const float* __restrict__ pre,
                                                               const float* __restrict__ u, int64_t ns, int plane, int cin,
                                                               int group_n, int64_t gs, float* __restrict__ gw, float* __restrict__ gb, int c_first) {
    __shared__ double scratch[4];
    const int c = c_first + blockIdx.x;
    const int grp = blockIdx.z / group_n, n = blockIdx.z - grp * group_n;
    const float* gp = gout + static_cast<int64_t>(blockIdx.z) * plane;
    const float* pp = pre + grp * gs + static_cast<int64_t>(n) * plane;
    const float* up = u + grp * gs + n * ns + static_cast<int64_t>(c < cin ? c : 0) * plane;
    float part = 0.f, part2 = 0.f;
    if ((plane & 3) == 0) {
        const int stride = gridDim.y * blockDim.x * 4;
        for (int i = (blockIdx.y * blockDim.x + threadIdx.x) * 4; i < plane; i += stride) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(gp + i);
            const f32x4 pv = *reinterpret_cast<const f32x4*>(pp + i);
            f32x4 uv = {1.f, 1.f, 1.f, 1.f};
            if (c < cin) uv = *reinterpret_cast<const f32x4*>(up + i);
            part += gv[0] * sign_of(pv[0]) * uv[0] + gv[1] * sign_of(pv[1]) * uv[1];
            part2 += gv[2] * sign_of(pv[2]) * uv[2] + gv[3] * sign_of(pv[3]) * uv[3];
        }
    } else {
        for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < plane; i += gridDim.y * blockDim.x)
            part += gp[i] * sign_of(pp[i]) * (c < cin ? up[i] : 1.f);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double v = wave_sum(static_cast<double>(part + part2));
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = static_cast<float>(scratch[0] + scratch[1] + scratch[2] + scratch[3]);
        if (c < cin) atomicAdd(gw + c, t); else atomicAdd(gb, t);
    }
}

// dW_final[c] += sum over the persistent base pass's blocks of its partial: one wave per channel, fixed order (fp64)
__global__ void __launch_bounds__(64) final_w_reduce_kernel(const double* __restrict__ parts, int nblocks, int count, float* __restrict__ gw) {
    const int c = blockIdx.x;
    double t = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) t += parts[static_cast<int64_t>(b) * count + c];
    t = wave_sum(t);
    if (threadIdx.x == 0) atomicAdd(gw + c, static_cast<float>(t));
}

// ---------------------------------------------------------------------------------------------
// schedule helpers
// ---------------------------------------------------------------------------------------------
struct Ctx {
    const endo_net* net;
    const float* params;
    float* bn_running;
    float* tape;
    float* grads;
    float* gradws;
    int training;
    hipStream_t stream;
    BiasParts* bias_parts = nullptr;

    int nt() const { return net->n * net->groups; }      // samples of all groups
    // context of the weight-gradient side stream, ordered after everything issued so far on the main stream
    int fork_wgrad(Ctx& side, int level) const {
        side = *this;
        if (!net->wstream || !net->opt[ENDO_OPT_WGRAD_OVERLAP] || level < kSideStreamFromLevel) return 0;          // run in line
        ENDO_CHECK(hipEventRecord(net->ev_fork, stream));
        ENDO_CHECK(hipStreamWaitEvent(net->wstream, net->ev_fork, 0));
        side.stream = net->wstream;
        return 0;
    }
    float* act(int level) const { return tape + net->lv[level].act; }
    float* gbuf(int level) const { return gradws + net->lv[level].grad; }
    double* sums(int level) const { return reinterpret_cast<double*>(reinterpret_cast<char*>(tape) + net->sums_off) + net->lv[level].sums; }
    // where a kernel accumulates the per-channel sum / sum^2 of what it stores: nowhere in inference mode (running statistics are
    // used, reference evaluate.py:317-345), which drops the reductions and their atomics from every epilogue
    double* out_sums(int level, int channel) const { return training ? sums(level) + 2 * channel : nullptr; }
    float* saved(const BnP& b) const { return tape + net->saved_off + 2 * b.saved; }
    double* scratch(const BnP& b) const { return reinterpret_cast<double*>(reinterpret_cast<char*>(gradws) + net->scratch_off) + 2 * b.saved; }
    float* pq_p(int level) const { return gradws + net->pq_off + 2 * net->lv[level].pq; }
    float* pq_q(int level) const { return pq_p(level) + net->lv[level].t; }
    uint8_t* idx(int level) const { return reinterpret_cast<uint8_t*>(tape) + net->idx_off[level]; }
};

static void fill_bn_in(const Ctx& c, ConvParams& p, const BnP& b, int level, int ic0) {
    const auto& lv = c.net->lv[level];
    p.in_sums = c.sums(level) + 2 * ic0;
    p.gamma = c.params + b.g;
    p.beta = c.params + b.b;
    p.running_mean = c.bn_running + b.run;
    p.running_var = c.bn_running + b.run + b.c;
    p.saved = c.tape ? c.saved(b) : nullptr;
    p.count = static_cast<double>(c.net->n) * lv.h * lv.w;      // per sample group
    p.eps = kBnEps;
    p.momentum = kBnMomentum;
    p.training = c.training;
}

static void fill_in(const Ctx& c, ConvParams& p, const float* base, int level, int ic0, int cin) {
    const auto& lv = c.net->lv[level];
    p.in = base + ic0 * lv.plane;
    p.in_ns = lv.t * lv.plane;
    p.in_cs = static_cast<int>(lv.plane);
    p.in_w = lv.w;
    p.cin = cin;
}

static void fill_out(const Ctx& c, ConvParams& p, float* base, int level, int oc0, int cout) {
    const auto& lv = c.net->lv[level];
    p.out = base + oc0 * lv.plane;
    p.out_ns = lv.t * lv.plane;
    p.out_cs = static_cast<int>(lv.plane);
    p.out_w = lv.w;
    p.cout = cout;
}

static void fill_grid(const Ctx& c, ConvParams& p, int level) {
    const auto& lv = c.net->lv[level];
    p.n = c.nt(); p.h = lv.h; p.w = lv.w;
    p.group_n = c.net->n; p.gs = c.net->gs; p.in_gs = c.net->gs; p.out_gs = c.net->gs;      // tape / workspace pointers by default
}

static double conv_flops(const endo_net* net, int level, int cin, int cout, int ks) {
    return 2.0 * net->n * net->groups * net->lv[level].plane * cin * cout * ks * ks;
}

// Tuning options, per network handle (endo_net_set_option; defaults set by endo_net_create*, no environment variables):
//   ENDO_OPT_WINO_FWD        dense-layer forward at the fine levels: 0 = direct convolution, 1 = Winograd F(2x2, 3x3) (2 LDS stages), 3 / 4 = the same
//                            with 3 / 4 stages, 5 (default) = F(4x4, 3x3) for the launches that fill the chip with 64 x 16 blocks, F(2x2, 3x3) for the rest
//   ENDO_OPT_WINO_DGRAD      fused base-channel data gradient at the fine levels: 0 = direct, 1 = Winograd, phase-skewed (dgrad_wino3_kernels.h),
//                            2 = Winograd, round-2 kernel (dgrad_wino_kernels.h), 3 (default) = 1 as persistent blocks where that form applies
//                            (dgrad_wino3p_kernels.h: at most 144 base channels), the per-tile kernel elsewhere
//   ENDO_OPT_DGRAD_VEC       new-channel passes: 2 (default) = persistent blocks (dgrad_newmap_kernels.h), 1 / 0 = one block per tile with 16-byte / dword
//                            DMA of the gradient tiles (dgrad_block_kernels.h)
//   ENDO_OPT_MFMA_BF16       1 = bf16 MFMA operands in the dense layers' kernels (a different function: DESIGN.md 4.10)
//   ENDO_OPT_WINO_MIN_TILES  a Winograd kernel is used from this many tiles per launch on (default 1024: the levels whose launches fill
//                            the chip several times; tests set 1 to reach the kernels at small sizes)
//   ENDO_OPT_MFMA_X3         bit mask of the kernel families (1 wgrad, 2 forward, 4 dgrad) that evaluate fp32 products as three-term bf16 splits (common.h)
//   ENDO_OPT_WGRAD_OVERLAP   1 = weight gradients on the side stream, 0 = in line on the caller's stream
//   ENDO_OPT_WGRAD_F34       1 = dense weight gradients of the fine levels in the Winograd domain F(3x3, 4x4) (wgrad_f34_kernels.h)
//   ENDO_OPT_FINAL_VIRTUAL   1 = the final convolution's data gradient is not written out: the last up block's kernels form g * w[c] (FinalVirt)
//   ENDO_OPT_TD_PERSIST      transition-down layers as persistent blocks: bit 0 the data gradient (td_dgrad_kernels.h), bit 1 the forward (td_fwd_kernels.h)
static void default_options(int (&opt)[ENDO_OPT_COUNT]) {
    opt[ENDO_OPT_WINO_FWD] = 5;          // F(4x4, 3x3) where its 64 x 16 blocks fill the chip (level 0 of configs[1]), F(2x2, 3x3) below: depth 5e-6 of its maximum from fp64 against the 1e-4 of the parity target
    opt[ENDO_OPT_WINO_DGRAD] = 3;
    opt[ENDO_OPT_DGRAD_VEC] = 2;
    opt[ENDO_OPT_WINO_MIN_TILES] = 1024;
    opt[ENDO_OPT_MFMA_BF16] = 0;
    opt[ENDO_OPT_WGRAD_OVERLAP] = 1;
    opt[ENDO_OPT_MFMA_X3] = 0;
    opt[ENDO_OPT_WGRAD_F34] = 1;
    opt[ENDO_OPT_FINAL_VIRTUAL] = 1;
    opt[ENDO_OPT_TD_PERSIST] = 3;
}
static int wino_fwd_mode(const Ctx& c) { return c.net->opt[ENDO_OPT_WINO_FWD]; }
static bool wino_fwd_enabled(const Ctx& c) { return wino_fwd_mode(c) != 0; }
static int wino_dgrad_mode(const Ctx& c) { return c.net->opt[ENDO_OPT_WINO_DGRAD]; }
static bool wino_dgrad_enabled(const Ctx& c) { return wino_dgrad_mode(c) != 0; }
static bool dgrad_vec_enabled(const Ctx& c) { return c.net->opt[ENDO_OPT_DGRAD_VEC] != 0; }
// bit 0: weight gradients, bit 1: forward, bit 2: data gradients of the dense layers (1 = all three)
static int mfma_bf16_mask(const Ctx& c) { const int v = c.net->opt[ENDO_OPT_MFMA_BF16]; return v == 1 ? 7 : (v >> 1); }
static bool mfma_bf16_wgrad(const Ctx& c) { return (mfma_bf16_mask(c) & 1) != 0; }
// operand mode of a dense-layer weight gradient: 0 fp32 MFMA, 1 operands rounded to bf16, 2 fp32 operands as three-term bf16 splits
static int wgrad_mfma_mode(const Ctx& c) { return mfma_bf16_wgrad(c) ? 1 : ((c.net->opt[ENDO_OPT_MFMA_X3] & 1) ? 2 : 0); }
static bool mfma_bf16_fwd(const Ctx& c) { return (mfma_bf16_mask(c) & 2) != 0; }
static bool mfma_bf16_dgrad(const Ctx& c) { return (mfma_bf16_mask(c) & 4) != 0; }

// dense layer forward: BN -> ReLU -> conv3x3 -> +12 channels (reference models.py:19-28, 44-52)
// fin_w / fin_pre / fused_final: the network's last dense layer may also form the final convolution's sum over its input channels
// (ConvParams::fin_w); *fused_final says whether the kernel form that ran did
// The tile-count part of dense_fwd's choice of a Winograd form at a level (the kernels' own shape checks can still send a layer to the direct
// kernel).  Where neither holds the level's dense layers run the direct kernel, and the forward pass prepares their weights in the K-chunk
// pipeline's order instead of the Winograd domain (WinoWeightTable::mode 1, ConvParams::wgt_chunks).
static bool dense_fwd_wino4_tiles(const Ctx& c, int level) {
    const auto& lv = c.net->lv[level];
    const long t4 = static_cast<long>((lv.w + 63) / 64) * ((lv.h + 15) / 16) * c.nt();
    return wino_fwd_mode(c) == 5 && !mfma_bf16_fwd(c) && 2 * t4 >= c.net->opt[ENDO_OPT_WINO_MIN_TILES];
}
static bool dense_fwd_wino2_tiles(const Ctx& c, int level) {
    const auto& lv = c.net->lv[level];
    const long t16 = static_cast<long>((lv.w + 31) / 32) * ((lv.h + 15) / 16) * c.nt();
    const long t8 = static_cast<long>((lv.w + 31) / 32) * ((lv.h + 7) / 8) * c.nt();
    const long min_tiles = c.net->opt[ENDO_OPT_WINO_MIN_TILES];
    return wino_fwd_enabled(c) && !mfma_bf16_fwd(c) && (t16 >= min_tiles || t8 >= (min_tiles * 3) / 4);
}
static bool dense_fwd_chunk_weights(const Ctx& c, int level) {
    return !mfma_bf16_fwd(c) && !dense_fwd_wino4_tiles(c, level) && !dense_fwd_wino2_tiles(c, level);
}

static int dense_fwd(const Ctx& c, int level, int ic0, int oc0, const BnP& b, const ConvP& cv, const float* fin_w = nullptr, float* fin_pre = nullptr,
                     bool* fused_final = nullptr) {
    const auto& lv = c.net->lv[level];
    ConvParams p{};
    fill_grid(c, p, level);
    fill_in(c, p, c.act(level), level, ic0, cv.cin);
    fill_bn_in(c, p, b, level, ic0);
    p.wgt = c.params + cv.w; p.bias = c.params + cv.b; p.w_cout = cv.cout; p.w_cin = cv.cin;
    fill_out(c, p, c.act(level), level, oc0, cv.cout);
    p.out_sums = c.out_sums(level, oc0);
    ProfScope prof(kProfConv3x3Dense, c.stream, conv_flops(c.net, level, cv.cin, cv.cout, 3),
                   4.0 * c.nt() * lv.plane * (cv.cin + cv.cout));
    // Fine levels: Winograd F(2x2, 3x3) on the matrix cores -- 4/9 of the multiply-accumulates (wino_fwd_kernels.h).
    // 32 x 16 pixel tiles while they fill the chip several times over, 32 x 8 below that.
    if (cv.u >= 0 && dense_fwd_wino4_tiles(c, level)) {          // F(4x4, 3x3): 36 instead of 64 products per 16 pixels (wino4_fwd_kernels.h)
        ConvParams p4 = p;
        p4.wgt = c.tape + c.net->wino4_off + cv.u / kWinoUStride * kW4UStride;
        if (wino4_fwd_ok(p4)) {          // (level 0 of configs[1]: at level 1 the 64 x 16 blocks no longer fill the chip, measured slower)
            if (fin_w && fused_final && c.net->opt[ENDO_OPT_FINAL_VIRTUAL]) { p4.fin_w = fin_w + ic0; p4.fin_out = fin_pre; *fused_final = true; }
            return launch_wino4_fwd(p4, c.stream);
        }
    }
    if (cv.u >= 0 && dense_fwd_wino2_tiles(c, level)) {
        ConvParams pw = p;
        pw.wgt = c.tape + c.net->wino_off + cv.u;          // group 0's tape: weights are shared by the groups
        const long t16 = static_cast<long>((lv.w + 31) / 32) * ((lv.h + 15) / 16) * c.nt();
        const long t8 = static_cast<long>((lv.w + 31) / 32) * ((lv.h + 7) / 8) * c.nt();
        if (wino_fwd_ok(pw)) {
            const int mode = wino_fwd_mode(c);          // in-job A/B: 1 = 2 LDS stages (default), 3 = 3 stages, 4 = 4 stages
            const long min_tiles = c.net->opt[ENDO_OPT_WINO_MIN_TILES];
            const bool big = t16 >= min_tiles, small = t8 >= (min_tiles * 3) / 4;
            if (mode == 4) {
                if (big) return launch_wino_fwd<2, 4, 2, 4>(pw, c.stream);
                if (small) return launch_wino_fwd<1, 4, 3, 4>(pw, c.stream);
            }
            if (mode == 3) {
                if (big) return launch_wino_fwd<2, 4, 2, 3>(pw, c.stream);
                if (small) return launch_wino_fwd<1, 4, 3, 3>(pw, c.stream);
            }
#ifdef ENDO_WINO_DIAG          // stub-out timing variants of the level-0 kernel (wrong results): -DENDO_WINO_DIAG, ENDO_WINO_EXP=<mask>
            if (big) {
                static const int exp = [] { const char* e = std::getenv("ENDO_WINO_EXP"); return e ? std::atoi(e) : 0; }();
                switch (exp) {
                    case 14: return launch_wino_fwd<2, 4, 2, 2, 14>(pw, c.stream);
                    case 30: return launch_wino_fwd<2, 4, 2, 2, 30>(pw, c.stream);
                    case 31: return launch_wino_fwd<2, 4, 2, 2, 31>(pw, c.stream);
                    case 63: return launch_wino_fwd<2, 4, 2, 2, 63>(pw, c.stream);
                    case 64: return launch_wino_fwd<2, 4, 2, 2, 64>(pw, c.stream);
                    default: break;
                }
            }
#endif
            if (big) return launch_wino_fwd<2, 4, 2, 2>(pw, c.stream);
            // (round 5, in-job A/B of the level-1 launch: K-chunks of 8 channels, 3 or 4 LDS stages, 4 blocks per CU -- all within +-0.2 % of this form)
            if (small) return launch_wino_fwd<1, 4, 3, 2>(pw, c.stream);
        }
    }
    // Coarse levels have too few 16x8 tiles to fill 256 CUs and a long K loop (Cin up to 372): slice K over
    // blockIdx.y, write raw partial sums, and let a small kernel add them up (+ bias, + BN statistics).
    // Scratch bound: slices * N * plane <= (768 / tiles + 1) * 128 * tiles <= 98304 + 65536 floats per channel.
    const long tiles_big = static_cast<long>((lv.w + 31) / 32) * ((lv.h + 15) / 16) * c.nt();
    const long tiles_mid = static_cast<long>((lv.w + 15) / 16) * ((lv.h + 15) / 16) * c.nt();
    const long tiles_small = static_cast<long>((lv.w + 15) / 16) * ((lv.h + 7) / 8) * c.nt();
    const int nchunks = (cv.cin + 15) / 16;
    // both launches below that run on K-chunks of 16 channels take the chunk-ordered copy of the weights the pass prepared for this level
    const float* chunk_weights = (cv.u >= 0 && dense_fwd_chunk_weights(c, level)) ? c.tape + c.net->wino_off + cv.u : nullptr;
    if (tiles_big < 512 && tiles_mid < 384 && tiles_small < 512 && nchunks >= 4) {
        int want = static_cast<int>(((tiles_small < 256 ? 512 : 768) + tiles_small - 1) / tiles_small);
        if (want > nchunks) want = nchunks;
        const int per = (nchunks + want - 1) / want;
        const int ksplit = (nchunks + per - 1) / per;
        float* partial = c.tape + c.net->partial_off;
        p.ksplit = ksplit;
        p.split_stride = static_cast<int64_t>(c.net->n) * cv.cout * lv.plane;      // inside one group's tape
        p.out = partial; p.out_ns = static_cast<int64_t>(cv.cout) * lv.plane;
        p.bias = nullptr; p.out_sums = nullptr;
        p.wgt_chunks = chunk_weights;
        int rc = mfma_bf16_fwd(c) ? launch_conv_dma<3, 16, 1, IN_BNRELU, EPI_FWD, 1, 2, 2, 1, 1>(p, c.stream)
                             : launch_conv_dma<3, 16, 1, IN_BNRELU, EPI_FWD, 1, 2, 2, 1>(p, c.stream);
        if (rc) return rc;
        int bx = static_cast<int>((lv.plane + 255) / 256);
        bx = bx > 8 ? 8 : bx;
        finalize_partial_kernel<<<dim3(bx, cv.cout, c.nt()), 256, 0, c.stream>>>(
            partial, p.split_stride, ksplit, p.out_ns, static_cast<int>(lv.plane), c.params + cv.b, c.act(level) + oc0 * lv.plane,
            lv.t * lv.plane, c.out_sums(level, oc0), c.net->n, c.net->gs);
        ENDO_LAUNCH_CHECK();
        return 0;
    }
    // Tile shape by block count (tools/conv_bench, Cin = 228 at 128x160): the launch wants >= ~1000 blocks.
    //   16 samples: 32x16 (640 blocks) 257 us, 32x8 (1280) 216 us;   8 samples: 16x16 (640) 143 us, 16x8 (1280) 125 us
    const long tiles_wide = static_cast<long>((lv.w + 31) / 32) * ((lv.h + 7) / 8) * c.nt();
    // level 0 stays on 32x16: in the training step (A/B of two library builds inside one job) it is 5 % faster than 32x8,
    // although the isolated microbenchmark prefers 32x8 by 5 %
    if (mfma_bf16_fwd(c)) {          // bf16 MFMA operands (ENDO_OPT_MFMA_BF16): the same tile shapes
        if (tiles_big < 1024 && tiles_wide >= 1024) return launch_conv_dma<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 4, 2, 1, 1>(p, c.stream);
        if (tiles_big < 1024 && tiles_small >= 768) return launch_conv_dma<3, 8, 1, IN_BNRELU, EPI_FWD, 1, 2, 2, 1, 1>(p, c.stream);
        return launch_conv_dma_auto<3, 4, 1, IN_BNRELU, EPI_FWD, 8, 2, 1, 1>(p, c.stream);
    }
    if (tiles_big < 1024 && tiles_wide >= 1024) return launch_conv_dma<3, 4, 1, IN_BNRELU, EPI_FWD, 2, 4, 2, 1>(p, c.stream);
    if (tiles_big < 1024 && tiles_small >= 768) return launch_conv_dma<3, 8, 1, IN_BNRELU, EPI_FWD, 1, 2, 2, 1>(p, c.stream);
    p.wgt_chunks = chunk_weights;          // (only the KC = 16 instantiation the auto choice ends in for few tiles reads it)
    return launch_conv_dma_auto<3, 4, 1, IN_BNRELU, EPI_FWD, 8, 2, 1>(p, c.stream);
}

// transition down: BN -> ReLU -> conv1x1 -> maxpool2 into the next level (models.py:56-67)
static int td_fwd(const Ctx& c, int level, const BnP& b, const ConvP& cv) {
    const int next = level + 1;
    const int oc0 = next < kLevels ? 48 : 0;
    ConvParams p{};
    fill_grid(c, p, level);
    fill_in(c, p, c.act(level), level, 48, cv.cin);
    fill_bn_in(c, p, b, level, 48);
    p.wgt = c.params + cv.w; p.bias = c.params + cv.b; p.w_cout = cv.cout; p.w_cin = cv.cin;
    fill_out(c, p, c.act(next), next, oc0, cv.cout);
    p.out_idx = c.idx(level);      // [n][cout][pooled plane] bytes, channel index relative to `out`
    p.idx_ns = static_cast<int64_t>(cv.cout) * c.net->lv[next].plane;
    p.out_sums = c.out_sums(next, oc0);
    ProfScope prof(kProfConv1x1Pool, c.stream, conv_flops(c.net, level, cv.cin, cv.cout, 1),
                   4.0 * c.nt() * c.net->lv[level].plane * (cv.cin + cv.cout / 4.0));
    // levels 0 / 1 of configs[1] (96 / 144 channels, whole 32 x 8 tiles): persistent blocks, weights LDS-resident, all output channels per tile (td_fwd_kernels.h)
    if ((c.net->opt[ENDO_OPT_TD_PERSIST] & 2) && !mfma_bf16_fwd(c) && c.training && td_fwd_ok(p)) return launch_td_fwd(p, c.net->cus, c.stream);
    if (mfma_bf16_fwd(c)) return launch_conv_dma_auto<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 4, 2, 1, 1>(p, c.stream);
    return launch_conv_dma_auto<1, 8, 3, IN_BNRELU, EPI_FWD_POOL, 4>(p, c.stream);    // 32x8 tiles: -6 % in the in-job A/B (Q = 6 was 10 % slower; round 5: K-chunks of 16 channels +-0, of 32 +40 % on the family, Q = 6 with 16 +14 %)
}

// transition up: nearest x2 -> conv3x3 48->48 into channels [0,48) of the finer level (models.py:70-80)
static int tu_fwd(const Ctx& c, int level, int src_level, int src_c0, const ConvP& cv) {
    const auto& sv = c.net->lv[src_level];
    ProfScope prof(kProfConv3x3Up, c.stream, conv_flops(c.net, level, cv.cin, cv.cout, 3),
                   4.0 * c.nt() * c.net->lv[level].plane * (cv.cin / 4.0 + cv.cout));
    if (sv.w % 4 == 0 && cv.cout == kNew && cv.cin % 4 == 0) {
        // sub-pixel form: the two row phases on the low-resolution grid, 4/9 of the MACs.  The tap-summed weights
        // go to the (idle) split-K scratch of group 0's tape; they are shared by all groups.
        float* w3 = c.tape + c.net->partial_off;
        tu_phase_weights_kernel<<<(2 * cv.cin * 384 + 255) / 256, 256, 0, c.stream>>>(c.params + cv.w, cv.cout, cv.cin, w3);
        ENDO_LAUNCH_CHECK();
        {   // ONE launch for both row phases (gridDim.y = 2, conv_dma_kernel<.., PH = 2>)
            ConvParams p{};
            fill_grid(c, p, src_level);                     // the launch runs over the low-resolution pixels
            fill_in(c, p, c.act(src_level), src_level, src_c0, cv.cin);
            p.wgt = w3; p.w_cout = 2 * cv.cout; p.w_cin = cv.cin;
            p.bias = c.params + cv.b;
            fill_out(c, p, c.act(level), level, 0, cv.cout);
            p.out_sums = c.out_sums(level, 0);
            return launch_conv_dma_vec<3, 4, 6, IN_PLAIN, EPI_FWD, 2, 4, 2, 1, 4, 0, 0, 2>(p, c.stream);
        }
    }
    ConvParams p{};
    fill_grid(c, p, level);
    fill_in(c, p, c.act(src_level), src_level, src_c0, cv.cin);
    p.wgt = c.params + cv.w; p.bias = c.params + cv.b; p.w_cout = cv.cout; p.w_cin = cv.cin;
    fill_out(c, p, c.act(level), level, 0, cv.cout);
    p.out_sums = c.out_sums(level, 0);
    return launch_conv_dma_auto<3, 4, 3, IN_UPSAMPLE, EPI_FWD>(p, c.stream);
}

// The final convolution's data gradient as the "virtual" content of the level-0 gradient buffer (DgradBlockParams::vg): vg = the plane
// g = grad_out * sign(pre) in the gradient workspace, vw = the 192 final-conv weights; base: the base-channel pass forms it too
// (otherwise final_bwd_data_kernel has materialised the block's base channels)
struct FinalVirt { const float* vg; const float* vw; bool base; bool base_w; float* gw; };          // base_w: the base pass also forms the final convolution's weight gradient of its channels (gw: that tensor's gradient)

static int prep_dy(const Ctx& c, int level, int c0, int count, float* bias_grad, const BnFin4* fin = nullptr, int nl = 0, const FinalVirt* fv = nullptr) {
    const auto& lv = c.net->lv[level];
    BnFin4 none{};
    int bx = static_cast<int>((lv.plane + 4095) / 4096);      // 16 pixels per thread
    bx = bx < 1 ? 1 : (bx > 32 ? 32 : bx);
    float* parts = nullptr;
    if (bias_grad && c.bias_parts) {          // a full table is an error, not a silent fall-back to the kernel's atomics (whose order is not fixed)
        BiasParts& bp = *c.bias_parts;
        const int64_t need = static_cast<int64_t>(count) * c.nt() * bx;
        if (bp.table.n >= kBiasPartLaunches || bp.used + need > c.net->bias_parts_floats) return ENDO_E_UNSUPPORTED;          // (sized from the network table: endo_net_create_grouped)
        parts = c.gradws + c.net->bias_parts_off + bp.used;
        bp.table.e[bp.table.n++] = {parts, bias_grad, count, c.nt() * bx};
        bp.used += need;
    }
    ProfScope prof(kProfSmall, c.stream, 0.0, 12.0 * c.nt() * lv.plane * count);
    prep_dy_kernel<<<dim3(bx, count, c.nt()), 256, 0, c.stream>>>(c.gbuf(level) + c0 * lv.plane, c.act(level) + c0 * lv.plane,
                                                                     lv.t * lv.plane, static_cast<int>(lv.plane), c.pq_p(level) + c0,
                                                                     c.pq_q(level) + c0, bias_grad, c.net->n, c.net->gs, fin ? *fin : none, fin ? nl : 0,
                                                                     static_cast<double>(c.net->n) * lv.h * lv.w, c.training, c.net->slot_stride,
                                                                     fv ? fv->vg : nullptr, fv ? fv->vw + c0 : nullptr, parts);
    ENDO_LAUNCH_CHECK();
    return 0;
}

// BN parameter gradients + deferred dx terms for channels [first, first + count) of the layer's input
// (ic0 = level-buffer channel of the layer's first input channel)
static int bn_finalize(const Ctx& c, const BnP& b, int level, int ic0, int first = 0, int count = -1) {
    const auto& lv = c.net->lv[level];
    if (count < 0) count = b.c - first;
    if (count <= 0) return 0;
    ProfScope prof(kProfSmall, c.stream, 0.0, 0.0);
    bn_bwd_finalize_kernel<<<dim3((count + 127) / 128, c.net->groups), 128, 0, c.stream>>>(c.scratch(b) + 2 * first, c.saved(b) + 2 * first, c.params + b.g + first,
                                                                       c.grads + b.g + first, c.grads + b.b + first,
                                                                       c.pq_p(level) + ic0 + first, c.pq_q(level) + ic0 + first, count,
                                                                       static_cast<double>(c.net->n) * lv.h * lv.w, c.training, c.net->gs, c.net->slot_stride);
    ENDO_LAUNCH_CHECK();
    return 0;
}

static void fill_wgrad_grid(const Ctx& c, WgradParams& p, int level) {
    const auto& lv = c.net->lv[level];
    p.n = c.nt(); p.h = lv.h; p.w = lv.w;
    p.group_n = c.net->n; p.gs = c.net->gs; p.in_gs = c.net->gs;
    p.tiles_x = (lv.w + kWgTileX - 1) / kWgTileX;
    p.tiles_y = (lv.h + kWgTileY - 1) / kWgTileY;
}

// batch / slice: the F(3x3, 4x4) form leaves its partial sums in scratch slice `slice` and its reduction to the caller's batched launch
static int dense_wgrad(const Ctx& c, int level, int ic0, int oc0, const BnP& b, const ConvP& cv, F34ReduceBatch* batch = nullptr, int slice = 0) {
    const auto& lv = c.net->lv[level];
    WgradParams p{};
    fill_wgrad_grid(c, p, level);
    p.in = c.act(level) + ic0 * lv.plane; p.in_ns = lv.t * lv.plane; p.in_cs = static_cast<int>(lv.plane); p.in_w = lv.w; p.cin = cv.cin;
    p.saved = c.saved(b); p.gamma = c.params + b.g; p.beta = c.params + b.b;
    p.dy = c.gbuf(level) + oc0 * lv.plane; p.dy_ns = lv.t * lv.plane; p.dy_cs = static_cast<int>(lv.plane); p.dy_w = lv.w; p.cout = cv.cout;
    p.dw = c.grads + cv.w;
    ProfScope prof(kProfWgradDense, c.stream, conv_flops(c.net, level, cv.cin, cv.cout, 3), 4.0 * c.nt() * lv.plane * (cv.cin + cv.cout));
    if (c.net->opt[ENDO_OPT_WGRAD_F34] && wgrad_mfma_mode(c) == 0 && wgrad_f34_ok(p, c.net->opt[ENDO_OPT_WINO_MIN_TILES] / 4l))          // from 256 tiles of 4 x 4 pixels: levels 0-4 of configs[1] (level 5 is 8 x 10)
        return launch_wgrad_f34(p, c.gradws + c.net->wg_scratch_off + (batch ? slice : 0) * kF34ScratchFloats, c.stream,
                                c.net->opt[ENDO_OPT_WGRAD_F34] == 2 ? 256 : kF34Blocks, batch);          // Winograd F(3x3, 4x4)
    if (wgrad_nsplit_ok(p)) {
        if (wgrad_mfma_mode(c) == 2) return launch_wgrad_x3(p, c.gradws + c.net->wg_scratch_off, c.stream);          // fp32 products as bf16 splits (wgrad_x3_kernels.h)
        return launch_wgrad_nsplit(p, c.gradws + c.net->wg_scratch_off, c.stream, wgrad_mfma_mode(c));
    }
    if (wgrad_taps_ok(p)) return mfma_bf16_wgrad(c) ? launch_wgrad_taps<12, IN_BNRELU, 1>(p, c.stream) : launch_wgrad_taps<12, IN_BNRELU>(p, c.stream);
    return launch_wgrad<3, 1, IN_BNRELU, DY_PLAIN>(p, c.stream);
}

// dense layer backward: bias grad + deferred-term fold, wgrad, dgrad fused with ReLU/BN backward
static int dense_bwd(const Ctx& c, int level, int ic0, int oc0, const BnP& b, const ConvP& cv, int acc_from) {
    const auto& lv = c.net->lv[level];
    int rc = prep_dy(c, level, oc0, cv.cout, c.grads + cv.b);
    if (rc) return rc;
    {
        Ctx cw;
        rc = c.fork_wgrad(cw, level);
        if (rc) return rc;
        rc = dense_wgrad(cw, level, ic0, oc0, b, cv);
        if (rc) return rc;
    }
    {
        ConvParams p{};
        fill_grid(c, p, level);
        fill_in(c, p, c.gbuf(level), level, oc0, cv.cout);
        p.wgt = c.params + cv.w; p.w_cout = cv.cout; p.w_cin = cv.cin;
        fill_out(c, p, c.gbuf(level), level, ic0, cv.cin);
        p.x = c.act(level) + ic0 * lv.plane; p.x_ns = lv.t * lv.plane; p.x_cs = static_cast<int>(lv.plane);
        p.bn_saved = c.saved(b); p.bn_gamma = c.params + b.g; p.bn_beta = c.params + b.b;
        p.bn_scratch = c.scratch(b); p.bn_slot_stride = c.net->slot_stride;
        p.acc_from = acc_from - ic0;
        ProfScope prof(kProfDgradDense, c.stream, conv_flops(c.net, level, cv.cin, cv.cout, 3), 4.0 * c.nt() * lv.plane * (3.0 * cv.cin + cv.cout));
        rc = launch_dgrad_dense_auto(p, c.stream);
        if (rc) return rc;
    }
    return bn_finalize(c, b, level, ic0);
}

// Backward of a whole dense block (4 layers) whose input ("base") is level-buffer channels [ic0, ic0 + c0)
// and whose 4 x 12 new maps follow at [ic0 + c0, ic0 + c0 + 48).  base_overwrite: the base channels have
// no gradient yet (first writer) instead of accumulating.
//   fine levels : per layer (last to first) prep + wgrad + a dgrad restricted to the NEW channels the layer
//                 reads (they carry the layer-to-layer dependency); then ONE fused dgrad for the base channels
//                 of all four layers (dgrad_block_kernels.h) -- 3x less HBM traffic than four full dgrads
//   coarse levels: the per-layer path (few tiles: parallelism comes from splitting channels over blocks)
// Which base-channel kernel the fused path takes (0 = dgrad_block8, 1 = phase-skewed Winograd, 2 = round-2 Winograd)
static int base_pass_form(const Ctx& c, int level, const DgradBlockParams& p, const ConvP* cv) {
    const auto& lv = c.net->lv[level];
    const long wtiles = static_cast<long>(lv.w / 32) * (lv.h / 8) * c.nt();
    if (mfma_bf16_dgrad(c)) return 0;
    if (wino_dgrad_enabled(c) && dgrad_wino_ok(p) && wtiles >= c.net->opt[ENDO_OPT_WINO_MIN_TILES] && cv[0].ud >= 0)
        return ((wino_dgrad_mode(c) == 1 || wino_dgrad_mode(c) == 3) && dgrad_wino3_ok(p)) ? 1 : 2;
    return 0;
}

// fv: the block's gradient-buffer range starts out as the final convolution's rank-one data gradient (the last up block, level 0), which
// is never written: the first touch of every channel forms it (FinalVirt)
static int dense_block_bwd(const Ctx& c, int level, int ic0, int c0, const BnP* bn, const ConvP* cv, bool base_overwrite, const FinalVirt* fv = nullptr) {
    const auto& lv = c.net->lv[level];
    const int new0 = ic0 + c0;
    DgradBlockParams probe{};
    probe.w = lv.w; probe.cs = static_cast<int>(lv.plane); probe.ns = lv.t * lv.plane;
    probe.x = c.act(level) + ic0 * lv.plane; probe.out = c.gbuf(level) + ic0 * lv.plane;
    const long tiles = static_cast<long>((lv.w + 31) / 32) * ((lv.h + 5) / 6) * c.nt();
    (void)tiles;          // few tiles: the fused kernel slices the channel groups over blockIdx.y
    if (!dgrad_block_ok(probe)) {
        for (int j = kLayers - 1; j >= 0; --j) {
            const int acc_from = (j == kLayers - 1 && base_overwrite) ? new0 : 0;
            int rc = dense_bwd(c, level, ic0, new0 + kGrowth * j, bn[j], cv[j], acc_from < ic0 ? ic0 : acc_from);
            if (rc) return rc;
        }
        return 0;
    }
    auto fill_common = [&](DgradBlockParams& p) {
        p.n = c.nt(); p.h = lv.h; p.w = lv.w;
        p.group_n = c.net->n; p.gs = c.net->gs; p.slot_stride = c.net->slot_stride;
        p.g_ns = lv.t * lv.plane; p.g_cs = static_cast<int>(lv.plane); p.g_w = lv.w;
        p.ns = lv.t * lv.plane; p.cs = static_cast<int>(lv.plane);
    };
    BnFin4 pending{};          // BN layers whose sums over the next prepared maps the last new-channel pass produced (folded into prep_dy)
    int pending_nl = 0;
    F34ReduceBatch red{};      // the block's F(3x3, 4x4) weight gradients reduce their partial sums in ONE launch behind the last of them
    for (int j = kLayers - 1; j >= 0; --j) {
        // (with fv: layer 3's maps have no gradient in the buffer yet -- prep_dy is their first writer; the maps of layers 2..0 were
        // written, from the virtual content, by the new-channel passes below)
        // (round 5, in-job A/B: a separate one-block finalize launch in front of a prologue-free prep_dy makes this kernel family 0.15 ms per step
        // faster and the step none: 18.15 against 18.15 ms -- the 33 extra launches cost what the prologues did)
        int rc = prep_dy(c, level, new0 + kGrowth * j, kGrowth, c.grads + cv[j].b, &pending, pending_nl, (fv && j == kLayers - 1) ? fv : nullptr);
        if (rc) return rc;
        // ENDO_OPT_WGRAD_OVERLAP 1: fork after every prep_dy; 2: ONE fork per dense block, after its last prep_dy (nothing rewrites a
        // prepared G before the join, so the four weight gradients may start late; 55 -> 22 event record / wait pairs per backward pass)
        const bool defer = c.net->opt[ENDO_OPT_WGRAD_OVERLAP] == 2;
        if (!defer || j == 0) {
            Ctx cw;
            rc = c.fork_wgrad(cw, level);
            if (rc) return rc;
            for (int jj = defer ? kLayers - 1 : j; jj >= j; --jj) {
                rc = dense_wgrad(cw, level, ic0, new0 + kGrowth * jj, bn[jj], cv[jj], &red, jj);
                if (rc) return rc;
            }
            if (j == 0 && red.count > 0) {          // (the side stream runs its launches in order: all four are ahead of this one)
                ProfScope prof(kProfSmall, cw.stream, 0.0, 0.0);          // (not a convolution launch: kept out of the weight-gradient family's per-launch figures)
                rc = launch_wgrad_f34_reduce_batch(red, cw.stream);
                if (rc) return rc;
            }
        }
        if (j > 0) {
            // Gradient into the 12 new maps of layer j-1 from ALL its consumers inside the block (layers j..3) in one pass:
            // the layer-to-layer dependency only needs G_j..G_3 final, and this way every new map is read (x) and
            // read-modify-written (gradient) once instead of once per consumer -- 180 instead of 252 plane passes per block.
            const int nl = kLayers - j;
            const int t0 = c0 + kGrowth * (j - 1);          // index of the target channels inside the consumers' inputs
            DgradBlockParams p{};
            fill_common(p);
            p.g = c.gbuf(level) + (new0 + kGrowth * j) * lv.plane;
            p.x = c.act(level) + (ic0 + t0) * lv.plane;
            p.out = c.gbuf(level) + (ic0 + t0) * lv.plane;
            p.count = kGrowth;
            p.acc_from = 0;          // a later consumer (next block / transition) wrote these maps first
            if (fv) { p.vg = fv->vg; p.vw = fv->vw + ic0 + t0; }          // ... or nobody: the final convolution's gradient is formed here
            p.w_ci_off = t0;
            BnFin4 a{};
            for (int l = 0; l < nl; ++l) {
                const BnP& b = bn[j + l];
                p.wgt[l] = c.params + cv[j + l].w; p.w_cin[l] = cv[j + l].cin;
                p.saved[l] = c.saved(b) + 2 * t0; p.gamma[l] = c.params + b.g + t0; p.beta[l] = c.params + b.b + t0;
                p.scratch[l] = c.scratch(b) + 2 * t0;
                a.scratch[l] = p.scratch[l]; a.saved[l] = p.saved[l]; a.gamma[l] = p.gamma[l];
                a.ggamma[l] = c.grads + b.g + t0; a.gbeta[l] = c.grads + b.b + t0;
            }
            {
                ProfScope prof(kProfDgradDense, c.stream, 2.0 * c.nt() * lv.plane * kGrowth * kGrowth * 9 * nl,
                               4.0 * c.nt() * lv.plane * (3.0 * kGrowth + kGrowth * nl));
                if (mfma_bf16_dgrad(c) && dgrad_block_vec_ok(p))
                    rc = nl == 1 ? launch_dgrad_block<1, 2, 3, 1, 0, 1, 4, 1>(p, c.stream)
                       : nl == 2 ? launch_dgrad_block<2, 2, 3, 1, 0, 1, 4, 1>(p, c.stream) : launch_dgrad_block<3, 2, 3, 1, 0, 1, 4, 1>(p, c.stream);
                else if (dgrad_block_vec_ok(p) && c.net->opt[ENDO_OPT_DGRAD_VEC] >= 2 && dgrad_newmap_ok(p))
                    rc = nl == 1 ? launch_dgrad_newmap<1>(p, c.stream) : nl == 2 ? launch_dgrad_newmap<2>(p, c.stream) : launch_dgrad_newmap<3>(p, c.stream);
                else if (dgrad_block_vec_ok(p) && dgrad_vec_enabled(c))
                    rc = nl == 1 ? launch_dgrad_block<1, 2, 3, 1, 0, 1, 4>(p, c.stream)
                       : nl == 2 ? launch_dgrad_block<2, 2, 3, 1, 0, 1, 4>(p, c.stream) : launch_dgrad_block<3, 2, 3, 1, 0, 1, 4>(p, c.stream);
                else
                    rc = nl == 1 ? launch_dgrad_block<1, 2, 3>(p, c.stream)
                       : nl == 2 ? launch_dgrad_block<2, 2, 3>(p, c.stream) : launch_dgrad_block<3, 2, 3>(p, c.stream);
                if (rc) return rc;
            }
            pending = a;          // consumed by the prep_dy of these 12 maps at the top of the next iteration
            pending_nl = nl;
        }
    }
    {   // base channels, all four layers in one pass
        DgradBlockParams p{};
        fill_common(p);
        p.g = c.gbuf(level) + new0 * lv.plane;
        p.x = c.act(level) + ic0 * lv.plane;
        p.out = c.gbuf(level) + ic0 * lv.plane;
        p.count = c0;
        p.acc_from = base_overwrite ? c0 : 0;
        p.w_ci_off = 0;
        for (int j = 0; j < kLayers; ++j) {
            p.wgt[j] = c.params + cv[j].w; p.w_cin[j] = cv[j].cin;
            p.saved[j] = c.saved(bn[j]); p.gamma[j] = c.params + bn[j].g; p.beta[j] = c.params + bn[j].b;
            p.scratch[j] = c.scratch(bn[j]);
        }
        ProfScope prof(kProfDgradDense, c.stream, 2.0 * c.nt() * lv.plane * c0 * kGrowth * 9 * kLayers,
                       4.0 * c.nt() * lv.plane * (3.0 * c0 + kGrowth * kLayers));
        int rc;
        const int form = base_pass_form(c, level, p, cv);
        if (fv && fv->base) {
            if (form != 1) return ENDO_E_BADARG;          // endo_net_bwd asked for a virtual base only where the phase-skewed kernel runs
            p.vg = fv->vg; p.vw = fv->vw + ic0;
        }
        if (mfma_bf16_dgrad(c)) {
            rc = launch_dgrad_block8<4, 1>(p, c.stream);
        } else if (form != 0) {
            // fine levels: Winograd F(2x2, 3x3), 48 instead of 108 MFMAs per 64 pixels and step (dgrad_wino_kernels.h)
            const float* ub = c.gradws + c.net->wd_off;
            const float* const u[4] = {ub + cv[0].ud, ub + cv[1].ud, ub + cv[2].ud, ub + cv[3].ud};
            // form 1: the phase-skewed kernel (its U layout; endo_net_bwd transforms the weights to match), 2: the round-2 kernel
            if (form == 1 && wino_dgrad_mode(c) == 3 && dgrad_wino3p_applies(p)) {
                // persistent blocks, one per CU; with fv->base_w they leave per-block partials of dW_final[ic0 .. ic0 + c0), added up here
                double* fwp = (fv && fv->base && fv->base_w) ? reinterpret_cast<double*>(c.gradws + c.net->fw_parts_off) : nullptr;
                int used = 0;
                rc = run_dgrad_wino3p_nl4(p, u, c.net->cus < kFwPartBlocks ? c.net->cus : kFwPartBlocks, fwp, &used, c.stream);
                if (rc == 0 && fwp) {
                    final_w_reduce_kernel<<<c0, 64, 0, c.stream>>>(fwp, used, c0, fv->gw + ic0);
                    ENDO_LAUNCH_CHECK();
                }
            } else {
                if (fv && fv->base && fv->base_w) return ENDO_E_BADARG;          // endo_net_bwd left these channels' final-conv weight gradient to the persistent kernel
                rc = form == 1 ? run_dgrad_wino3_nl4(p, u, c.stream) : launch_dgrad_wino8<4>(p, u, c.stream);
            }
        } else {
            rc = launch_dgrad_block8<4>(p, c.stream);       // 512-thread blocks: +15 % over the 4-wave kernel (tools/conv_bench)
        }
        if (rc) return rc;
    }
    {
        BnFin4 a;
        for (int j = 0; j < kLayers; ++j) {
            a.scratch[j] = c.scratch(bn[j]); a.saved[j] = c.saved(bn[j]); a.gamma[j] = c.params + bn[j].g;
            a.ggamma[j] = c.grads + bn[j].g; a.gbeta[j] = c.grads + bn[j].b;
        }
        ProfScope prof(kProfSmall, c.stream, 0.0, 0.0);
        bn_bwd_finalize4_kernel<<<dim3((c0 + 127) / 128, c.net->groups), 128, 0, c.stream>>>(a, kLayers, c.pq_p(level) + ic0, c.pq_q(level) + ic0, c0,
                                                                         static_cast<double>(c.net->n) * lv.h * lv.w, c.training, c.net->gs,
                                                                         c.net->slot_stride);
        ENDO_LAUNCH_CHECK();
    }
    return 0;
}

static int td_bwd(const Ctx& c, int level, const BnP& b, const ConvP& cv) {
    const int next = level + 1;
    const int oc0 = next < kLevels ? 48 : 0;
    const auto& lv = c.net->lv[level];
    const auto& nx = c.net->lv[next];
    int rc = prep_dy(c, next, oc0, cv.cout, c.grads + cv.b);
    if (rc) return rc;
    {
        WgradParams p{};
        fill_wgrad_grid(c, p, level);
        p.in = c.act(level) + 48 * lv.plane; p.in_ns = lv.t * lv.plane; p.in_cs = static_cast<int>(lv.plane); p.in_w = lv.w; p.cin = cv.cin;
        p.saved = c.saved(b); p.gamma = c.params + b.g; p.beta = c.params + b.b;
        p.dy = c.gbuf(next) + oc0 * nx.plane; p.dy_ns = nx.t * nx.plane; p.dy_cs = static_cast<int>(nx.plane); p.dy_w = nx.w; p.cout = cv.cout;
        p.dy_idx = c.idx(level); p.idx_ns = static_cast<int64_t>(cv.cout) * nx.plane;
        p.dw = c.grads + cv.w;
        Ctx cw;
        rc = c.fork_wgrad(cw, level);
        if (rc) return rc;
        ProfScope prof(kProfWgradOther, cw.stream, conv_flops(c.net, level, cv.cin, cv.cout, 1), 4.0 * c.nt() * lv.plane * cv.cin);
        rc = wgrad1x1_dma_ok(p) ? (mfma_bf16_wgrad(c) ? launch_wgrad1x1_dma<1>(p, cw.stream) : launch_wgrad1x1_dma<0>(p, cw.stream))
                                : launch_wgrad1x1(p, cw.stream);
        if (rc) return rc;
    }
    {
        ConvParams p{};
        fill_grid(c, p, level);
        fill_in(c, p, c.gbuf(next), next, oc0, cv.cout);
        p.in_idx = c.idx(level); p.idx_ns = static_cast<int64_t>(cv.cout) * nx.plane;   // channel index relative to p.in
        p.wgt = c.params + cv.w; p.w_cout = cv.cout; p.w_cin = cv.cin;
        fill_out(c, p, c.gbuf(level), level, 48, cv.cin);
        p.x = c.act(level) + 48 * lv.plane; p.x_ns = lv.t * lv.plane; p.x_cs = static_cast<int>(lv.plane);
        p.bn_saved = c.saved(b); p.bn_gamma = c.params + b.g; p.bn_beta = c.params + b.b;
        p.bn_scratch = c.scratch(b); p.bn_slot_stride = c.net->slot_stride;
        p.acc_from = 0;
        ProfScope prof(kProfDgradOther, c.stream, conv_flops(c.net, level, cv.cin, cv.cout, 1), 4.0 * c.nt() * lv.plane * 3.0 * cv.cin);
        // pooled rows of whole code dwords -> LDS-DMA kernel; otherwise the register-staged one
        // levels 0 / 1 of configs[1] (96 / 144 channels, whole 32 x 8 tiles): persistent blocks, weights LDS-resident, 16-byte DMA (td_dgrad_kernels.h)
        if ((c.net->opt[ENDO_OPT_TD_PERSIST] & 1) && !mfma_bf16_dgrad(c) && td_dgrad_ok(p))
            rc = launch_td_dgrad(p, c.net->cus, c.stream);
        else if ((c.net->opt[ENDO_OPT_TD_PERSIST] & 1) && !mfma_bf16_dgrad(c) && nx.w % 4 != 0 && td_dgrad_small_ok(p))
            // pooled rows without whole code dwords (level 4 of configs[1]: 8 x 10): 128-pixel runs, the routed gradient expanded on its way into
            // LDS -- 46 instead of the register-staged kernel's 100 us.  (At levels 2 / 3 the LDS-DMA kernel stays: 103 / 63 against 139 / 69 us, tools/td_bench)
            rc = launch_td_dgrad_small(p, c.stream);
        else
        rc = (nx.w % 4 == 0) ? (mfma_bf16_dgrad(c) ? launch_conv_dma_auto<1, 16, 2, IN_UNPOOL, EPI_DGRAD_BN, 4, 2, 1, 1>(p, c.stream)
                                                  : launch_conv_dma_auto<1, 16, 2, IN_UNPOOL, EPI_DGRAD_BN, 4>(p, c.stream))
                             : launch_conv_auto<1, 16, 3, IN_UNPOOL, EPI_DGRAD_BN, 4>(p, c.stream);
        if (rc) return rc;
    }
    return bn_finalize(c, b, level, 48);
}

static int tu_bwd(const Ctx& c, int level, int src_level, int src_c0, const ConvP& cv) {
    const auto& lv = c.net->lv[level];
    const auto& sv = c.net->lv[src_level];
    int rc = prep_dy(c, level, 0, cv.cout, c.grads + cv.b);
    if (rc) return rc;
    {
        WgradParams p{};
        fill_wgrad_grid(c, p, level);
        p.in = c.act(src_level) + src_c0 * sv.plane; p.in_ns = sv.t * sv.plane; p.in_cs = static_cast<int>(sv.plane); p.in_w = sv.w; p.cin = cv.cin;
        p.dy = c.gbuf(level); p.dy_ns = lv.t * lv.plane; p.dy_cs = static_cast<int>(lv.plane); p.dy_w = lv.w; p.cout = cv.cout;
        p.dw = c.grads + cv.w;
        Ctx cw;
        rc = c.fork_wgrad(cw, level);
        if (rc) return rc;
        ProfScope prof(kProfWgradOther, cw.stream, conv_flops(c.net, level, cv.cin, cv.cout, 3), 4.0 * c.nt() * lv.plane * (cv.cin / 4.0 + cv.cout));
        WgradParams ps = p;                    // sub-pixel form walks the low-resolution grid
        ps.h = sv.h; ps.w = sv.w;
        if (tu_wgrad_subpix_ok(ps)) rc = launch_tu_wgrad_subpix(ps, c.gradws + c.net->wg_scratch_off, cw.stream);
        else rc = wgrad_taps_ok(p, true) ? launch_wgrad_taps<12, IN_UPSAMPLE>(p, cw.stream) : launch_wgrad<3, 1, IN_UPSAMPLE, DY_PLAIN>(p, cw.stream);
        if (rc) return rc;
    }
    ProfScope prof(kProfDgradOther, c.stream, conv_flops(c.net, level, cv.cin, cv.cout, 3), 4.0 * c.nt() * lv.plane * (cv.cout + cv.cin / 4.0));
    if (sv.w % 4 == 0 && cv.cin == kNew && cv.cout == kNew) {
        // sub-pixel form on the low-resolution grid: the four stride-2 phases of dY are 4 x 48 pseudo input channels of a
        // 3x3 convolution that uses 2x2 of its taps per phase (4/9 of the MACs); the weights have their own scratch (the n-split scratch belongs to the side stream)
        float* wd = c.gradws + c.net->tuw_scratch_off;
        tu_subpix_dgrad_weights_kernel<<<(16 * cv.cout * cv.cin + 255) / 256, 256, 0, c.stream>>>(c.params + cv.w, cv.cout, cv.cin, wd);
        ENDO_LAUNCH_CHECK();
        ConvParams p{};
        fill_grid(c, p, src_level);
        fill_in(c, p, c.gbuf(level), level, 0, 4 * cv.cout);      // strides of the full-resolution gradient buffer
        p.sub_c = cv.cout;
        p.wgt = wd; p.w_cout = cv.cin; p.w_cin = 4 * cv.cout;
        fill_out(c, p, c.gbuf(src_level), src_level, src_c0, cv.cin);
        // Tile shape by block count (round 6): on 32 x 8 tiles the launch of level 4 is 32 blocks and that of level 3 128 -- each walking all 24 K-chunks,
        // 93 and 103 us for 0.1 and 0.4 GFLOP.  16 x 8 / 16 x 4 tiles where 32 x 8 ones leave the chip under-filled.
        const long t32 = static_cast<long>((sv.w + 31) / 32) * ((sv.h + 7) / 8) * c.nt();
        const long t16 = static_cast<long>((sv.w + 15) / 16) * ((sv.h + 7) / 8) * c.nt();
        if (t32 >= 1024) return launch_conv_dma_vec<3, 8, 3, IN_SUBPIX, EPI_FWD, 2, 4, 2, 1, 1>(p, c.stream);
        if (t16 >= 512) return launch_conv_dma_vec<3, 8, 3, IN_SUBPIX, EPI_FWD, 1, 2, 2, 1, 1>(p, c.stream);
        return launch_conv_dma_vec<3, 8, 3, IN_SUBPIX, EPI_FWD, 1, 1, 2, 1, 1>(p, c.stream);
    }
    ConvParams p{};
    fill_grid(c, p, level);
    fill_in(c, p, c.gbuf(level), level, 0, cv.cout);
    p.wgt = c.params + cv.w; p.w_cout = cv.cout; p.w_cin = cv.cin;
    fill_out(c, p, c.gbuf(src_level), src_level, src_c0, cv.cin);
    return launch_conv_dma_auto<3, 4, 3, IN_PLAIN, EPI_DGRAD_SUMPOOL, 4>(p, c.stream);
}

static int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

}  // namespace endo

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int endo_net_create_grouped(endo_net** out, int n, int h, int w, int groups) {
    if (!out || n <= 0 || h <= 0 || w <= 0 || groups <= 0) return ENDO_E_BADARG;
    if (groups > kMaxGroups) return ENDO_E_UNSUPPORTED;
    if ((h % 32) != 0 || (w % 32) != 0) return ENDO_E_UNSUPPORTED;   // 5 poolings + exact centre crop (models.py:93-97)
    const Table& tb = table();
    endo_net* net = new (std::nothrow) endo_net();
    if (!net) return ENDO_E_BADARG;
    net->n = n; net->h = h; net->w = w; net->groups = groups;
    net->wstream = nullptr; net->ev_fork = nullptr; net->ev_join = nullptr;
    default_options(net->opt);
    int64_t off = 0, sums = 0, pq = 0;
    for (int l = 0; l <= kLevels; ++l) {
        auto& lv = net->lv[l];
        lv.h = h >> l; lv.w = w >> l; lv.t = level_channels(l);
        lv.plane = static_cast<int64_t>(lv.h) * lv.w;
        lv.act = off; lv.grad = off;
        off += align_up(static_cast<int64_t>(n) * lv.t * lv.plane, 64);
        lv.sums = sums; sums += 2 * lv.t;
        lv.pq = pq; pq += lv.t;
    }
    const int64_t acts = off;
    net->pre_off = off; off += align_up(static_cast<int64_t>(n) * h * w, 64);
    net->saved_off = off; off += align_up(2 * tb.bn_width_total, 64);
    int64_t byte_off = off * 4;
    for (int l = 0; l < kLevels; ++l) {
        net->idx_off[l] = byte_off;
        byte_off += align_up(static_cast<int64_t>(n) * (down_in(l) + kNew) * net->lv[l + 1].plane, 256);
    }
    net->sums_off = byte_off;
    net->sums_bytes = sums * 8;
    byte_off += align_up(net->sums_bytes, 256);
    net->partial_off = byte_off / 4;
    byte_off += static_cast<int64_t>(kGrowth) * 163840 * 4;     // bound: see dense_fwd
    net->wino_off = byte_off / 4;
    byte_off += align_up(tb.wino_floats * 4, 256);
    net->wino4_off = byte_off / 4;
    byte_off += align_up(tb.wino4_floats * 4, 256);
    net->tape_floats = byte_off / 4;
    net->pq_off = acts;
    net->pq_floats = 2 * pq;
    net->scratch_off = align_up((acts + net->pq_floats) * 4, 256);
    net->slot_stride = align_up(tb.bn_width_total * 2, 32);
    net->scratch_bytes = net->slot_stride * 8 * kBnSlots;
    net->wg_scratch_off = (net->scratch_off + align_up(net->scratch_bytes, 256)) / 4;
    net->tuw_scratch_off = net->wg_scratch_off + std::max(std::max(kNsScratchFloats, kSpScratchFloats), 4 * kF34ScratchFloats);          // four slices: the layers of a dense block reduce in one launch
    net->wd_off = align_up(net->tuw_scratch_off + 4 * kNew * (4 * kNew + 16), 64);
    net->gplane_off = align_up(net->wd_off + tb.wino_dgrad_floats, 64);
    net->bias_parts_off = net->gplane_off + align_up(static_cast<int64_t>(n) * h * w, 64);
    net->bias_parts_floats = static_cast<int64_t>(kBiasPartChannels) * n * groups * 32;
    {   // the BiasParts limits against THIS network's table: one prep_dy launch per convolution with a bias, its output channels in all
        int launches = 1, channels = tb.first.cout;
        for (int l = 0; l < kLevels; ++l) {
            for (int j = 0; j < kLayers; ++j) { launches += 2; channels += tb.down_conv[l][j].cout + tb.up_conv[l][j].cout; }
            launches += 2; channels += tb.td_conv[l].cout + tb.tu_conv[l].cout;
        }
        for (int j = 0; j < kLayers; ++j) { ++launches; channels += tb.bott_conv[j].cout; }
        if (launches > kBiasPartLaunches || channels > kBiasPartChannels) { delete net; return ENDO_E_UNSUPPORTED; }
    }
    net->fw_parts_off = net->bias_parts_off + align_up(net->bias_parts_floats, 64);
    net->gradws_floats = net->fw_parts_off + 2 * static_cast<int64_t>(kFwPartBlocks) * 192;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        net->cus = cus;
    }
    // one stride for both buffers keeps the kernels' group arithmetic to a single number; the caller allocates
    // groups * gs floats for each when groups > 1 (the two sizes differ by a few per cent)
    net->gs = align_up(net->tape_floats > net->gradws_floats ? net->tape_floats : net->gradws_floats, 64);
    *out = net;
    return 0;
}

extern "C" int endo_net_create(endo_net** out, int n, int h, int w) { return endo_net_create_grouped(out, n, h, w, 1); }

extern "C" int endo_net_set_option(endo_net* net, int option_id, int value) {
    if (!net || option_id < 0 || option_id >= ENDO_OPT_COUNT) return ENDO_E_BADARG;
    const int old = net->opt[option_id];
    net->opt[option_id] = value;
    return old;
}

extern "C" int endo_net_get_option(const endo_net* net, int option_id) {
    if (!net || option_id < 0 || option_id >= ENDO_OPT_COUNT) return ENDO_E_BADARG;
    return net->opt[option_id];
}

extern "C" void endo_net_destroy(endo_net* net) {
    if (!net) return;
    if (net->ev_fork) (void)hipEventDestroy(net->ev_fork);
    if (net->ev_join) (void)hipEventDestroy(net->ev_join);
    if (net->wstream) (void)hipStreamDestroy(net->wstream);
    delete net;
}
extern "C" int64_t endo_net_param_floats(void) { return table().param_floats; }
extern "C" int64_t endo_net_bn_floats(void) { return table().bn_floats; }
extern "C" int64_t endo_net_tape_floats(const endo_net* net) { return !net ? 0 : (net->groups > 1 ? net->groups * net->gs : net->tape_floats); }
extern "C" int64_t endo_net_gradws_floats(const endo_net* net) { return !net ? 0 : (net->groups > 1 ? net->groups * net->gs : net->gradws_floats); }
extern "C" int endo_net_groups(const endo_net* net) { return net ? net->groups : 0; }
extern "C" int64_t endo_net_group_stride(const endo_net* net) { return net ? net->gs : 0; }
extern "C" int64_t endo_net_param_offset(int index) {
    const Table& tb = table();
    if (index < 0 || index >= static_cast<int>(tb.param_offsets.size())) return -1;
    return tb.param_offsets[index];
}
extern "C" int64_t endo_net_bn_offset(int bn_index, int which) {
    const Table& tb = table();
    if (bn_index < 0 || bn_index >= static_cast<int>(tb.bn_order.size()) || which < 0 || which > 1) return -1;
    return tb.bn_order[bn_index]->run + which * tb.bn_order[bn_index]->c;
}
extern "C" int endo_net_level_channels(int level) { return (level < 0 || level > kLevels) ? -1 : level_channels(level); }
extern "C" int64_t endo_net_act_offset(const endo_net* net, int level) { return (!net || level < 0 || level > kLevels) ? -1 : net->lv[level].act; }

extern "C" int64_t endo_net_tape_offset(const endo_net* net, int what, int index) {
    if (!net) return -1;
    const Table& tb = table();
    switch (what) {
        case ENDO_TAPE_PRE: return net->pre_off;
        case ENDO_TAPE_BN_SAVED:
            if (index < 0 || index >= static_cast<int>(tb.bn_order.size())) return -1;
            return net->saved_off + 2 * tb.bn_order[index]->saved;
        case ENDO_TAPE_POOL:
            if (index < 0 || index >= kLevels) return -1;
            return net->idx_off[index];
        default: return -1;
    }
}

extern "C" int endo_net_fwd(endo_net* net, const float* params, float* bn_running, const float* x, float* out, float* tape,
                            int training, void* stream_) {
    if (!net || !params || !bn_running || !x || !out || !tape) return ENDO_E_BADARG;
    const Table& tb = table();
    Ctx c{net, params, bn_running, tape, nullptr, nullptr, training, static_cast<hipStream_t>(stream_)};
    for (int g = 0; g < net->groups; ++g)
        ENDO_CHECK(hipMemsetAsync(reinterpret_cast<char*>(tape + g * net->gs) + net->sums_off, 0, net->sums_bytes, c.stream));
    if (!mfma_bf16_fwd(c)) {          // dense-layer weights in Winograd form or in the direct kernel's chunk order, all 44 layers in one launch
        ProfScope prof(kProfSmall, c.stream, 0.0, 4.0 * (tb.wino_floats + tb.wino_floats * 9 / 16));
        WinoWeightTable wt = tb.wino;          // per pass: which layers want their weights in the direct kernel's chunk order (table order: down, bottleneck, up)
        for (int l = 0; l < wt.layers; ++l) {
            const int level = l < kLevels * kLayers ? l / kLayers : (l < (kLevels + 1) * kLayers ? kLevels : kLevels - 1 - (l - (kLevels + 1) * kLayers) / kLayers);
            wt.mode[l] = dense_fwd_chunk_weights(c, level) ? 1 : 0;
        }
        wino_fwd_weights_kernel<<<(wt.start[wt.layers] + 255) / 256, 256, 0, c.stream>>>(wt, params, tape + net->wino_off);
        ENDO_LAUNCH_CHECK();
        if (wino_fwd_mode(c) == 5) wino4_fwd_weights_kernel<<<(tb.wino4.start[tb.wino4.layers] + 255) / 256, 256, 0, c.stream>>>(tb.wino4, params, tape + net->wino4_off);
        ENDO_LAUNCH_CHECK();
    }
    int rc;
    bool fused_final = false;
    {   // first conv 3 -> 48 into level-0 channels [48, 96)
        ConvParams p{};
        fill_grid(c, p, 0);
        p.in = x; p.in_ns = 3 * net->lv[0].plane; p.in_cs = static_cast<int>(net->lv[0].plane); p.in_w = net->w; p.cin = 3;
        p.in_gs = net->n * p.in_ns;                 // x is the caller's [groups * n][3][H][W] tensor
        p.wgt = params + tb.first.w; p.bias = params + tb.first.b; p.w_cout = kFirst; p.w_cin = 3;
        fill_out(c, p, c.act(0), 0, 48, kFirst);
        p.out_sums = c.out_sums(0, 48);
        ProfScope prof(kProfConvFirst, c.stream, conv_flops(net, 0, 3, kFirst, 3), 4.0 * c.nt() * net->lv[0].plane * (3 + kFirst));
        rc = launch_conv_dma_auto<3, 4, 3, IN_PLAIN, EPI_FWD>(p, c.stream);
        if (rc) return rc;
    }
    for (int l = 0; l < kLevels; ++l) {
        for (int j = 0; j < kLayers; ++j) {
            rc = dense_fwd(c, l, 48, 48 + down_in(l) + kGrowth * j, tb.down_bn[l][j], tb.down_conv[l][j]);
            if (rc) return rc;
        }
        rc = td_fwd(c, l, tb.td_bn[l], tb.td_conv[l]);
        if (rc) return rc;
    }
    for (int j = 0; j < kLayers; ++j) {
        rc = dense_fwd(c, kLevels, 0, 288 + kGrowth * j, tb.bott_bn[j], tb.bott_conv[j]);
        if (rc) return rc;
    }
    for (int i = 0; i < kLevels; ++i) {
        const int l = kLevels - 1 - i;
        const int src = l + 1;
        const int src_c0 = (i == 0) ? 288 : 96 + down_in(src);
        rc = tu_fwd(c, l, src, src_c0, tb.tu_conv[i]);
        if (rc) return rc;
        for (int j = 0; j < kLayers; ++j) {
            const bool last = l == 0 && j == kLayers - 1;          // the layer whose 180 input channels are all but 12 of the final convolution's
            rc = dense_fwd(c, l, 0, 96 + down_in(l) + kGrowth * j, tb.up_bn[i][j], tb.up_conv[i][j], last ? params + tb.final_.w : nullptr,
                           last ? tape + net->pre_off : nullptr, last ? &fused_final : nullptr);
            if (rc) return rc;
        }
    }
    {
        const auto& lv = net->lv[0];
        const int c_first = fused_final ? 96 + down_in(0) + kGrowth * (kLayers - 1) : 0;          // 180: the last layer's launch has summed channels [0, 180) into `pre`
        ProfScope prof(kProfConvFinal, c.stream, 2.0 * c.nt() * lv.plane * 192, 4.0 * c.nt() * lv.plane * (194 - c_first));
        int bx = static_cast<int>((lv.plane / 4 + 255) / 256);
        bx = bx < 1 ? 1 : bx;
        final_fwd_kernel<<<dim3(bx, c.nt()), 256, 0, c.stream>>>(c.act(0), lv.t * lv.plane, static_cast<int>(lv.plane), 192,
                                                                 params + tb.final_.w, params + tb.final_.b, tape + net->pre_off, out,
                                                                 net->n, net->gs, c_first);
        ENDO_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int endo_net_bwd(endo_net* net, const float* params, const float* x, const float* tape, const float* grad_out,
                            float* grads, float* gradws, int training, void* stream_) {
    if (!net || !params || !x || !tape || !grad_out || !grads || !gradws) return ENDO_E_BADARG;
    const Table& tb = table();
    Ctx c{net, params, nullptr, const_cast<float*>(tape), grads, gradws, training, static_cast<hipStream_t>(stream_)};
    BiasParts bias_parts{};
    c.bias_parts = &bias_parts;
    if (!net->wstream) {          // side stream of the weight gradients (see endo_net), created on first use on the caller's device
        ENDO_CHECK(hipStreamCreateWithFlags(&net->wstream, hipStreamNonBlocking));
        ENDO_CHECK(hipEventCreateWithFlags(&net->ev_fork, hipEventDisableTiming));
        ENDO_CHECK(hipEventCreateWithFlags(&net->ev_join, hipEventDisableTiming));
    }
    // zero the deferred-term tables and the BN reduction scratch
    for (int g = 0; g < net->groups; ++g)
        ENDO_CHECK(hipMemsetAsync(gradws + g * net->gs + net->pq_off, 0,
                                  static_cast<size_t>(net->scratch_off + net->scratch_bytes - net->pq_off * 4), c.stream));
    if (wino_dgrad_enabled(c) && !mfma_bf16_dgrad(c)) {          // data-gradient weights of the dense layers in Winograd form, one launch
        ProfScope prof(kProfSmall, c.stream, 0.0, 4.0 * 2.0 * tb.wino_dgrad_floats);
        // mode 1: blocks the phase-skewed kernel takes (dgrad_wino3_ok: at most 12 base-channel groups) get its U layout
        dgrad_wino_weights_kernel<<<(tb.wino_dgrad.start[tb.wino_dgrad.layers] + 255) / 256, 256, 0, c.stream>>>(
            tb.wino_dgrad, params, gradws + net->wd_off, (wino_dgrad_mode(c) == 1 || wino_dgrad_mode(c) == 3) ? DgradWino3Geom<4>::kMaxCount / 16 : 0);
        ENDO_LAUNCH_CHECK();
    }
    int rc;
    FinalVirt virt{};
    bool use_virt = false;
    {
        const auto& lv = net->lv[0];
        // The final convolution's data gradient is rank one: dX[c] = g * w[c], g = grad_out * sign(pre).  Writing it out (192 planes, 1 GB
        // at 16 x 256 x 320) only for the last up block to read it back costs two passes over the level-0 buffer; instead g goes to one
        // plane and that block's kernels form the products where they first touch a channel (FinalVirt) -- where the block takes the
        // fused path, and for its base channels where the phase-skewed Winograd kernel runs; what is left is materialised as before.
        DgradBlockParams probe{};
        probe.n = c.nt(); probe.h = lv.h; probe.w = lv.w; probe.count = 96 + down_in(0);
        probe.cs = static_cast<int>(lv.plane); probe.ns = lv.t * lv.plane;
        probe.g_cs = static_cast<int>(lv.plane);
        probe.x = c.act(0); probe.out = c.gbuf(0);
        virt.vg = gradws + net->gplane_off; virt.vw = params + tb.final_.w; virt.base = false; virt.base_w = false; virt.gw = grads + tb.final_.w;
        int materialise = 192;          // channels [0, materialise) are written by final_bwd_data_kernel
        if (c.net->opt[ENDO_OPT_FINAL_VIRTUAL] && dgrad_block_ok(probe)) {
            use_virt = true;
            virt.base = base_pass_form(c, 0, probe, tb.up_conv[kLevels - 1]) == 1;
            materialise = virt.base ? 0 : 96 + down_in(0);
            // ... and where that kernel runs as persistent blocks it also forms sum g * x[c] of the base channels it streams: the final
            // convolution's weight gradient of those channels (round 6: final_bwd_weight_kernel then reads 48 + 1 instead of 192 + 1 planes)
            virt.base_w = virt.base && wino_dgrad_mode(c) == 3 && dgrad_wino3p_applies(probe);
        }
        {   // final conv weight / bias gradient: reads only grad_out and the tape, so it goes to the side stream first
            Ctx cw;
            rc = c.fork_wgrad(cw, 0);
            if (rc) return rc;
            const int c_first = virt.base_w ? 96 + down_in(0) : 0;
            int by = static_cast<int>((lv.plane + 256 * 16 - 1) / (256 * 16));       // 16 pixels per thread
            by = by < 1 ? 1 : (by > 16 ? 16 : by);
            ProfScope prof(kProfConvFinal, cw.stream, 2.0 * c.nt() * lv.plane * (192 - c_first), 4.0 * c.nt() * lv.plane * (192 - c_first + 2));
            final_bwd_weight_kernel<<<dim3(192 - c_first + 1, by, c.nt()), 256, 0, cw.stream>>>(grad_out, tape + net->pre_off, c.act(0), lv.t * lv.plane,
                                                                         static_cast<int>(lv.plane), 192, net->n, net->gs, grads + tb.final_.w,
                                                                         grads + tb.final_.b, c_first);
            ENDO_LAUNCH_CHECK();
        }
        ProfScope prof(kProfConvFinal, c.stream, 2.0 * c.nt() * lv.plane * 192, 4.0 * c.nt() * lv.plane * (materialise + 2));
        if (use_virt) {
            int bx = static_cast<int>((lv.plane + 1023) / 1024);
            final_g_kernel<<<dim3(bx, c.nt()), 256, 0, c.stream>>>(grad_out, tape + net->pre_off, gradws + net->gplane_off, static_cast<int>(lv.plane), net->n, net->gs);
            ENDO_LAUNCH_CHECK();
        }
        if (materialise > 0) {
            int bx = static_cast<int>((lv.plane + 255) / 256);
            final_bwd_data_kernel<<<dim3(bx, c.nt()), 256, 0, c.stream>>>(grad_out, tape + net->pre_off, params + tb.final_.w, c.gbuf(0),
                                                                          lv.t * lv.plane, static_cast<int>(lv.plane), materialise, net->n, net->gs, 0);
            ENDO_LAUNCH_CHECK();
        }
    }
    for (int i = kLevels - 1; i >= 0; --i) {
        const int l = kLevels - 1 - i;
        rc = dense_block_bwd(c, l, 0, 96 + down_in(l), tb.up_bn[i], tb.up_conv[i], l > 0, (l == 0 && use_virt) ? &virt : nullptr);
        if (rc) return rc;
        const int src = l + 1;
        const int src_c0 = (i == 0) ? 288 : 96 + down_in(src);
        rc = tu_bwd(c, l, src, src_c0, tb.tu_conv[i]);
        if (rc) return rc;
    }
    rc = dense_block_bwd(c, kLevels, 0, 288, tb.bott_bn, tb.bott_conv, true);
    if (rc) return rc;
    for (int l = kLevels - 1; l >= 0; --l) {
        rc = td_bwd(c, l, tb.td_bn[l], tb.td_conv[l]);
        if (rc) return rc;
        rc = dense_block_bwd(c, l, 48, down_in(l), tb.down_bn[l], tb.down_conv[l], false);
        if (rc) return rc;
    }
    {   // first conv: bias grad + weight grad (the image needs no gradient)
        const auto& lv = net->lv[0];
        WgradParams p{};
        fill_wgrad_grid(c, p, 0);
        p.in = x; p.in_ns = 3 * lv.plane; p.in_cs = static_cast<int>(lv.plane); p.in_w = lv.w; p.cin = 3;
        p.in_gs = net->n * p.in_ns;                 // the caller's image tensor
        p.dy = c.gbuf(0) + 48 * lv.plane; p.dy_ns = lv.t * lv.plane; p.dy_cs = static_cast<int>(lv.plane); p.dy_w = lv.w; p.cout = kFirst;
        p.dw = grads + tb.first.w;
        // The F(3x3, 4x4) form prepares the gradient itself (G = d + P x + Q, bias gradient = sum G: WgradParams::prep_x): this is the LAST kernel
        // of the backward pass, nothing else is on the chip, and prep_dy's own pass over 3 x 48 planes would be 0.1 ms of the step
        const bool f34 = c.net->opt[ENDO_OPT_WGRAD_F34] && wgrad_mfma_mode(c) == 0 && wgrad_f34_raw_ok(p, c.net->opt[ENDO_OPT_WINO_MIN_TILES] / 4l);
        const bool fuse_prep = f34 && c.net->opt[ENDO_OPT_FINAL_VIRTUAL];
        if (fuse_prep) {
            p.prep_x = c.act(0) + 48 * lv.plane;
            p.prep_p = c.pq_p(0) + 48; p.prep_q = c.pq_q(0) + 48;
            p.prep_bias = grads + tb.first.b;
        } else {
            rc = prep_dy(c, 0, 48, kFirst, grads + tb.first.b);
            if (rc) return rc;
        }
        Ctx cw;
        rc = c.fork_wgrad(cw, 0);
        if (rc) return rc;
        {
            ProfScope prof(kProfWgradOther, cw.stream, conv_flops(net, 0, 3, kFirst, 3), 4.0 * c.nt() * lv.plane * (3 + kFirst));
            // 3 -> 48 channels: in the F(3x3, 4x4) form the four sets of 12 output channels share one launch (the tap-folded kernel runs a
            // 108-row GEMM with 3 of 16 columns in use, four times: 216 us alone on the chip at the very end of the backward)
            if (fuse_prep)
                rc = launch_wgrad_f34<0, true, true>(p, c.gradws + c.net->wg_scratch_off, cw.stream);
            else if (f34)
                rc = launch_wgrad_f34<0, true>(p, c.gradws + c.net->wg_scratch_off, cw.stream);
            else
            rc = wgrad_taps_ok(p) ? launch_wgrad_taps<12, IN_PLAIN>(p, cw.stream) : launch_wgrad<3, 3, IN_PLAIN, DY_PLAIN>(p, cw.stream);
            if (rc) return rc;
        }
    }
    if (bias_parts.table.n > 0) {          // the conv-bias gradients from prep_dy's per-block sums (BiasParts): one launch for the whole pass
        ProfScope prof(kProfSmall, c.stream, 0.0, 4.0 * bias_parts.used);
        bias_reduce_kernel<<<dim3(bias_parts.table.n, 8), 256, 0, c.stream>>>(bias_parts.table);
        ENDO_LAUNCH_CHECK();
    }
    if (net->wstream) {          // join: the caller's stream continues only after every weight gradient has landed
        ENDO_CHECK(hipEventRecord(net->ev_join, net->wstream));
        ENDO_CHECK(hipStreamWaitEvent(c.stream, net->ev_join, 0));
    }
    return 0;
}
