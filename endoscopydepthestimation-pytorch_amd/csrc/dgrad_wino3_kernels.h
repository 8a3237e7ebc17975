// Fused data gradient of a dense block's base channels in Winograd F(2x2, 3x3) form, phase-skewed (round 3).
//
// Same arithmetic as dgrad_wino8_kernel (dgrad_wino_kernels.h) -- per step (one 16-channel group of the block's input, one layer)
//     dX[tile][channel] = A^T [ sum_c U_xi[c][channel] .* V_xi[tile][c] ] A ,   c = the layer's 12 prepared dY maps
// -- restructured so that the matrix pipe and the vector ALU of a SIMD work at the same time.  fp32 MFMA and VALU are separate
// pipes (MI355X_MICROARCH.md "Wave scheduling"; tools/mfma_rate_probe: an MFMA + two VALU per wave at two waves per SIMD take the
// MFMA's time, not the sum), but a wave issues in order, so only ANOTHER wave can fill the pipe a wave is not using.  In the
// round-2 kernel all 8 waves of the one resident block left the per-step barrier in the same phase: transform (VALU) together,
// MFMAs together, epilogue (VALU) together -- the SQ counters showed MFMA busy 0.27, VALU busy 0.30, waiting 0.42.
//
// Here a step of a wave is cut into two phases of about equal length,
//     V_s = E_{s-1} + T_s : output transform / ReLU mask / BN backward of the previous step, then the input transform B^T d B of
//                           this step's 3 x 4 dY maps into 48 registers                                   (~330 VALU, no MFMA)
//     M_s                 : 48 MFMAs (16 transform-domain GEMMs x 3 channel quads), B operands by ds_read_b128   (no VALU)
// and the block's two workers (4 waves = the 4 tile rows each; wave i and wave i + 4 share a SIMD) run HALF A STEP APART: in every
// interval between two block barriers one worker is in its M phase and the other in its V phase.  Everything a phase waits for
// was issued at least one interval earlier:
//   * the weight slice of step s + 1 (12 KB, LDS-DMA) at the start of M_s, waited for at the end of V_{s+1};
//   * x of the next channel group / the old gradient of a group's last layer at the start of the V phase before the one that
//     uses them, drained by that phase's closing wait;
//   * a finished group's result is parked in registers and stored at the start of the next M phase;
//   * the BN-backward sums go to a per-(worker, step, wave) LDS table and become fp64 atomics once, after the last interval
//     (the round-2 kernel issued them after every barrier and the next s_waitcnt vmcnt(0) waited for them).
// The BN constants (scale, beta, mean, rstd) of all NL layers sit in an LDS table, one ds_read_b128 per step.
// Work split: groups alternate between the workers; with an odd group count the last group's layers are split between them
// (worker 0 layers [0, NL/2), worker 1 the rest) and worker 1 hands its partial sum over through LDS at the end -- both workers
// always run the same number of steps.
//
// U layout (dgrad_wino_weights_kernel, layout 1): [group][c 12][a 4][j 16][i 4] with xi = 4 a + i -- a lane's four B values of a
// transform row a are one conflict-free ds_read_b128 (16 consecutive j = 16 consecutive 16-byte slots).
// LDS: dY tile 66 KB + U 2 workers x 2 buffers x 12 KB + sums 24 KB + BN table 12 KB = 150 KB: one 8-wave block per CU.
#pragma once

#include "dgrad_wino_kernels.h"

namespace endo {

template <int NL>
struct DgradWino3Geom {
    static_assert(NL % 2 == 0, "the split of an odd group count halves the layers");
    static constexpr int kThreads = 512;
    static constexpr int kTileX = 32, kTileY = 8;
    static constexpr int kRows = kTileY + 2, kCols = kTileX + 2;
    static constexpr int kPlane = kRows * kCols;                        // 340
    static constexpr int kCS = 352;                                     // == 32 (mod 64) dwords: the 8-byte patch reads of the 4 maps of a quad hit disjoint banks
    static constexpr int kU = kWinoDgradSlice;
    static constexpr int kMaxCount = 192;                               // base channels (12 groups: the level-1 up block)
    static constexpr int kMaxSteps = (kMaxCount / 32) * NL;             // per worker
    static constexpr int kRedStep = 4 * 16 * 2;                         // [4 waves][16][2]
    static constexpr int kG = NL * 12 * kCS;
    static constexpr size_t kBytes = sizeof(float) * (kG + 2 * 2 * kU + 2 * kMaxSteps * kRedStep + NL * kMaxCount * 4);
    static_assert(kBytes <= 160 * 1024, "one block per CU");
};

// p.w % 32 == 0, p.h % 8 == 0, p.count % 16 == 0, 32 <= p.count <= 192; u[l]: the layer's transformed weights (layout 1), group-major.
// EXP: diagnostic bit mask for tools/wino_bench (0 in the library; timing only): 1 = no x / gradient loads, 2 = no stores,
// 4 = no BN-sum atomics, 8 = no dY tile load, 32 = no MFMAs
// OPT (in-job A/B, tools/wino_bench; 0 in the library): 16 = weight DMA issued at the END of the V phases with a counted wait (M phases
// then carry no memory instruction), 32 = s_setprio 1 around the MFMAs of an M phase.  Both measured neutral to slightly slower.
template <int NL, int EXP = 0, int OPT = 0>
__global__ void __launch_bounds__(512, 2) dgrad_wino3_kernel(const DgradBlockParams p, const float* __restrict__ u0, const float* __restrict__ u1,
                                                             const float* __restrict__ u2, const float* __restrict__ u3) {
    using G = DgradWino3Geom<NL>;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int grp = p.group_n > 0 ? blockIdx.z / p.group_n : 0;
    const int n = blockIdx.z - grp * p.group_n;
    const int64_t grp_off = grp * p.gs;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_g = smem;                               // [NL*12][kCS]
    float* s_u = s_g + G::kG;                        // [worker][2][c 12][a 4][j 16][4]
    float* s_red = s_u + 4 * G::kU;                  // [worker][step][4 waves][16][2]
    float* s_bn = s_red + 2 * G::kMaxSteps * G::kRedStep;          // [NL][count][scale, beta, mean, rstd]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 2, w4 = wave & 3;         // worker; tile row of the wave
    const int th = tid & 255;
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int tile = (gridDim.x & 7) == 0 ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int x0 = (tile % p.tiles_x) * G::kTileX;
    const int y0 = (tile / p.tiles_x) * G::kTileY;
    const int px = x0 + 8 * lk;                      // the lane's 8 output columns (tiles 4 lk .. 4 lk + 3)
    const int py = y0 + 2 * w4;                      // and its 2 output rows
    const int ngroups = p.count / 16;
    const int nfull = ngroups >> 1;                  // whole groups per worker
    const bool odd = (ngroups & 1) != 0;
    const int nsteps = nfull * NL + (odd ? NL / 2 : 0);
    const float* const u_layer[4] = {u0, u1, u2, u3};

    // The worker's step list: its whole groups wk, wk + 2, ... (NL layers each), then -- odd group count -- NL / 2 layers of the last group.
    const int split_l0 = wk * (NL / 2);              // first layer of this worker's part of a split group

    // ---- prologue: dY tile (NL*12 maps with a 1-pixel halo, dword DMA by all 8 waves), BN table, first weight slices, first x ----
    {
        int goff = 0;
        bool ok = false;
        if (tid < G::kPlane) {
            const int ry = tid / G::kCols, rx = tid - ry * G::kCols;
            const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
            if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) { ok = true; goff = gy * p.g_w + gx; }
        }
        const float* g_n = p.g + grp_off + n * p.g_ns;
        const int e0 = wave * 64;
        if (e0 < G::kPlane && !(EXP & 8)) {
            for (int c = 0; c < NL * 12; ++c) {
                const float* src = ok ? g_n + static_cast<int64_t>(c) * p.g_cs + goff : g_pad_consts + 4;
                if (e0 + lane < G::kPlane) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(s_g + c * G::kCS + e0), 4, 0, 0);
            }
        }
    }
    for (int i = tid; i < NL * p.count; i += G::kThreads) {
        const int l = i / p.count, ch = i - l * p.count;
        const float mean = p.saved[l][grp_off + 2 * ch], rstd = p.saved[l][grp_off + 2 * ch + 1];
        *reinterpret_cast<f32x4*>(s_bn + 4 * i) = f32x4{p.gamma[l][ch] * rstd, p.beta[l][ch], mean, rstd};
    }
    // this worker's U slice of (group, layer) into weight buffer `buf`: one contiguous 12 KB run, 3 float4 units per thread
    auto issue_weights = [&](int gq, int l, int buf) {
        const float* src = u_layer[l] + static_cast<int64_t>(gq) * G::kU + 4 * th;
        float* dst = s_u + (wk * 2 + buf) * G::kU + w4 * 256;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + k * 1024), (lptr_t)(dst + k * 1024), 16, 0, 0);
    };
    const float* x_n = p.x + grp_off + n * p.ns;
    float* out_n = p.out + grp_off + n * p.ns;
    // 32-bit element offsets from the sample's base (a sample's channel range is far below 2^31 floats): SGPR base + VGPR offset addressing
    const unsigned pix0 = static_cast<unsigned>(py * p.w + px);
    auto pix_off = [&](int co, int r, int hh) { return static_cast<unsigned>(co) * static_cast<unsigned>(p.cs) + pix0 + static_cast<unsigned>(r * p.w + 4 * hh); };
    auto load_x = [&](int co, f32x4 (&dst)[2][2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if constexpr ((EXP & 1) != 0) dst[r][hh] = f32x4{0.1f * lane, 0.2f, -0.3f, 0.4f};
                else dst[r][hh] = *reinterpret_cast<const f32x4*>(x_n + pix_off(co, r, hh));
            }
    };
    // first touch of the next group's x one V phase ahead (one dword per 16-byte unit: 4 registers instead of 16), so that the real
    // loads -- issued after the epilogue has finished with the old group's x -- come from L2 and are covered by the transform
    auto touch_x = [&](int co, float (&t)[4]) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) t[2 * r + hh] = (EXP & 1) ? 0.f : x_n[pix_off(co, r, hh)];
    };
    // the old gradient of a group: the buffer's content -- or, with p.vg (dgrad_block_kernels.h: the final convolution's rank-one data
    // gradient, never materialised), g of the lane's pixels (the same 64 bytes for every group: L1 / L2 hits) times the channel's
    // final-conv weight
    const float* vg_n = p.vg ? p.vg + grp_off + static_cast<int64_t>(n) * p.cs : nullptr;
    auto load_old = [&](int co, f32x4 (&dst)[2][2]) {
        const float wf = vg_n ? p.vw[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if ((EXP & 1) != 0) dst[r][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
                else if (vg_n) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(vg_n + pix0 + static_cast<unsigned>(r * p.w + 4 * hh));
                    dst[r][hh] = f32x4{gv[0] * wf, gv[1] * wf, gv[2] * wf, gv[3] * wf};
                } else if (co < p.acc_from) dst[r][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
                else dst[r][hh] = *reinterpret_cast<const f32x4*>(out_n + pix_off(co, r, hh));
            }
    };
    auto store_out = [&](int co, const f32x4 (&src)[2][2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if constexpr ((EXP & 2) != 0) asm volatile("" ::"v"(src[r][hh][0]), "v"(src[r][hh][1]), "v"(src[r][hh][2]), "v"(src[r][hh][3]));
                else *reinterpret_cast<f32x4*>(out_n + pix_off(co, r, hh)) = src[r][hh];
            }
    };

    f32x4 xc[2][2], dc[2][2], total[2][2];           // [row][column half]: 8 consecutive pixels of 2 rows
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) { total[r][hh] = f32x4{0.f, 0.f, 0.f, 0.f}; dc[r][hh] = total[r][hh]; }
    // this step's A operands: V_xi of the lane's patch of map 4 quad + lk, xi = 4 a + i at av[quad][2 a + (i >> 1)][i & 1] -- register PAIRS: the
    // transforms are written on pairs (v_pk_add_f32 / v_pk_fma_f32).  That does NOT make them faster on gfx950 -- a packed fp32 instruction
    // takes two issue slots, measured round 5: 45 % fewer vector instructions, the same time -- but the compiler allocates this form in
    // 216 instead of 254 registers, which is the room the virtual old gradient (load_old) needs
    f32x2 av[3][8];
    f32x4 acc[16];

    // ---- T: input transform of the lane's 4x4 patches of layer l's 12 maps: raw patch values into av (24 8-byte reads), transformed in place ----
    auto load_patches = [&](int l) {
#pragma unroll
        for (int quad = 0; quad < 3; ++quad) {
            // LDS rows 2 w4 .. 2 w4 + 3, columns 2 li .. 2 li + 3 (two aligned pairs): row r's pairs land in av[quad][2 r], av[quad][2 r + 1]
            const float* a_base = s_g + (l * 12 + quad * 4 + lk) * G::kCS + (2 * w4) * G::kCols + 2 * li;
#pragma unroll
            for (int row = 0; row < 4; ++row) {
                av[quad][2 * row] = *reinterpret_cast<const f32x2*>(a_base + row * G::kCols);
                av[quad][2 * row + 1] = *reinterpret_cast<const f32x2*>(a_base + row * G::kCols + 2);
            }
        }
    };
    auto transform_inplace = [&]() {
        if constexpr ((EXP & 64) != 0) return;
#pragma unroll
        for (int quad = 0; quad < 3; ++quad) {
            // B^T d on the column pairs (c0, c1) | (c2, c3): rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3) -- 8 packed adds
            // (there is no packed subtract and clang scalarises a - b on ext-vectors: the difference is a packed add with the neg bits set;
            // operands here come from LDS reads, never straight from an MFMA -- the hazard recogniser does not look into inline assembly)
            auto sub2 = [](const f32x2 a, const f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; };
            const f32x2 l0 = av[quad][0], h0 = av[quad][1], l1 = av[quad][2], h1 = av[quad][3];
            const f32x2 l2 = av[quad][4], h2 = av[quad][5], l3 = av[quad][6], h3 = av[quad][7];
            const f32x2 tl[4] = {sub2(l0, l2), l1 + l2, sub2(l2, l1), sub2(l1, l3)};
            const f32x2 th[4] = {sub2(h0, h2), h1 + h2, sub2(h2, h1), sub2(h1, h3)};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                // (.) B: (v0, v1) = (c0 - c2, c1 + c2), (v2, v3) = (c2 - c1, c1 - c3) -- one packed add each: the operand-select bits pick c2 for
                // both halves / c1 for both halves, the neg bits the signs (clang turns the same arithmetic on ext-vectors into moves + xor)
                f32x2 v01, v23;
                asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(v01) : "v"(tl[a]), "v"(th[a]));
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v23) : "v"(th[a]), "v"(tl[a]));
                av[quad][2 * a] = v01;
                av[quad][2 * a + 1] = v23;
            }
        }
    };
    auto transform = [&](int l) { load_patches(l); transform_inplace(); };

    // ---- M: 16 transform-domain GEMMs over the layer's 12 dY maps: M = this wave's row of 16 tiles, N = the worker's 16 channels ----
    auto mfmas = [&](int buf) {
        if constexpr ((EXP & 128) != 0) {          // diagnostic: no M phase at all
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = f32x4{av[0][i >> 1][i & 1], av[1][i >> 1][i & 1], av[2][i >> 1][i & 1], av[0][i >> 1][i & 1]};
            return;
        }
        if constexpr ((OPT & 32) != 0) __builtin_amdgcn_s_setprio(1);
        const float* ub = s_u + (wk * 2 + buf) * G::kU;
#pragma unroll
        for (int quad = 0; quad < 3; ++quad) {
            const float* b_base = ub + ((quad * 4 + lk) * 4 * 16 + li) * 4;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(b_base + a * 64);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr ((EXP & 32) != 0) {
                        if (quad == 0) acc[4 * a + i] = f32x4{av[quad][2 * a + (i >> 1)][i & 1] * b[i], 0.f, 0.f, 0.f};
                        else acc[4 * a + i][quad] += av[quad][2 * a + (i >> 1)][i & 1] * b[i];
                    } else if (quad == 0) {
                        acc[4 * a + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[quad][2 * a + (i >> 1)][i & 1], b[i], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    } else {
                        acc[4 * a + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[quad][2 * a + (i >> 1)][i & 1], b[i], acc[4 * a + i], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr ((OPT & 32) != 0) __builtin_amdgcn_s_setprio(0);
    };

    // ---- E: output transform A^T M A (tiles 4 lk + e), layer l's ReLU mask + BN backward, accumulated over the layers;
    //         `slot` = the worker's running step number (its row of the BN-sum table) ----
    auto read_bn = [&](int gq, int l) { return *reinterpret_cast<const f32x4*>(s_bn + 4 * (l * p.count + gq * 16 + li)); };
    auto epilogue = [&](int gq, int l, int slot, const f32x4 bn) {
        if constexpr ((EXP & 64) != 0) {          // diagnostic: no V-phase arithmetic
#pragma unroll
            for (int i = 0; i < 16; ++i) total[i >> 3][(i >> 2) & 1][i & 3] += acc[i][0];
            return;
        }
        // On register pairs: the output transform A^T M A runs on the tile pairs (e, e + 1) of the accumulators (adjacent registers of an MFMA
        // result), BN + ReLU backward on the pixel pairs (k, k + 1) of x / the running total (adjacent registers of a 16-byte load); the
        // per-pixel select between the two (dz = z > 0 ? d : 0) writes its results where the second layout wants them: the transposition.
        const f32x2 sb = {bn[0], bn[1]}, mr = {bn[2], bn[3]};          // (scale, beta), (mean, rstd)
        const float rstd = bn[3];
        f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
        f32x2 minus1 = {-1.f, -1.f};
        asm("" : "+v"(minus1));          // opaque: with a visible constant the fma below folds back into a subtraction, which is scalarised
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {          // tiles e = 2 hh, 2 hh + 1 = the lane's column half hh
            f32x2 u0r[4], u1r[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x2 m0 = {acc[c][2 * hh], acc[c][2 * hh + 1]}, m1 = {acc[4 + c][2 * hh], acc[4 + c][2 * hh + 1]};
                const f32x2 m2 = {acc[8 + c][2 * hh], acc[8 + c][2 * hh + 1]}, m3 = {acc[12 + c][2 * hh], acc[12 + c][2 * hh + 1]};
                u0r[c] = m0 + m1 + m2;
                u1r[c] = __builtin_elementwise_fma(m2 + m3, minus1, m1);          // m1 - m2 - m3 (compiler-made instructions here: they read MFMA results, and the MFMA -> VALU wait states are the compiler's to insert)
            }
            // d[r][cx] over the tile pair: pixel k = 2 (e & 1) + cx of the column half
            const f32x2 d[2][2] = {{u0r[0] + u0r[1] + u0r[2], __builtin_elementwise_fma(u0r[2] + u0r[3], minus1, u0r[1])},
                                   {u1r[0] + u1r[1] + u1r[2], __builtin_elementwise_fma(u1r[2] + u1r[3], minus1, u1r[1])}};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int ee = 0; ee < 2; ++ee) {          // pixels k = 2 ee, 2 ee + 1
                    const f32x2 x2 = {xc[r][hh][2 * ee], xc[r][hh][2 * ee + 1]};
                    f32x2 xcen, z;
                    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(xcen) : "v"(x2), "v"(mr));              // x - mean
                    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(z) : "v"(xcen), "v"(sb));                     // xcen * scale + beta
                    const f32x2 dz = {z[0] > 0.f ? d[r][0][ee] : 0.f, z[1] > 0.f ? d[r][1][ee] : 0.f};
                    s1v += dz;
                    s2v = __builtin_elementwise_fma(dz, xcen, s2v);
                    f32x2 t2 = {total[r][hh][2 * ee], total[r][hh][2 * ee + 1]};
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(t2) : "v"(dz), "v"(sb));                                    // total += dz * scale
                    total[r][hh][2 * ee] = t2[0];
                    total[r][hh][2 * ee + 1] = t2[1];
                }
            }
        }
        float s1 = s1v[0] + s1v[1], s2 = s2v[0] + s2v[1];
        s2 *= rstd;
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
        if (lk == 0) {
            float* red = s_red + ((wk * G::kMaxSteps + slot) * 4 + w4) * 32;
            *reinterpret_cast<f32x2*>(red + 2 * li) = f32x2{s1, s2};
        }
    };
    // closing wait of a V phase that has just issued the 3 DMA instructions of a later weight slice: everything older has landed
    // (vector memory reads retire in order; the parked stores were issued at the START of the phase, ahead of every load)
    auto phase_end_v3 = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0073);          // vmcnt(3) lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // a phase ends at a block barrier; the two workers run one phase apart (worker 1 starts one barrier late, worker 0 ends one late)
    auto phase_end_v = [&]() {
        // the builtin, not inline assembly: the compiler's own wait-count bookkeeping sees it and adds no second vmcnt(0) where
        // xc / dc are first used (which would also wait for the loads a LATER phase has just issued)
        __builtin_amdgcn_s_waitcnt(0x0070);          // vmcnt(0) lgkmcnt(0): weight slice landed, loads back, LDS writes done
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto phase_end_m = [&]() {
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0) only: the DMA of the next slice stays in flight across the barrier
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    const int g_first = wk;                          // p.count >= 32: every worker has at least one whole group
    issue_weights(g_first, 0, 0);
    load_x(g_first * 16 + li, xc);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    if (wk == 1) __builtin_amdgcn_s_barrier();          // the skew
    transform(0);
    if constexpr ((OPT & 16) != 0) {
        issue_weights(g_first, 1, 1);
        phase_end_v3();
    } else {
        phase_end_v();
    }

    int slot = 0;
    for (int gi = 0; gi < nfull; ++gi) {
        const int gq = 2 * gi + wk;
        const int co = gq * 16 + li;
        // the step after this group's last: the next whole group, this worker's part of the split group, or nothing
        const bool has_next = gi + 1 < nfull || odd;
        const int g_next = gi + 1 < nfull ? gq + 2 : ngroups - 1;
        const int l_next = gi + 1 < nfull ? 0 : split_l0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            // ---------------- M(gq, l): parked stores, next weight slice, 48 MFMAs ----------------
            constexpr bool kLate = (OPT & 16) != 0;          // weight DMA at the end of the V phases, counted wait
            if (!kLate && l == 0 && gi > 0) store_out(co - 32, dc);
            if constexpr (!kLate) {
                if (l + 1 < NL) issue_weights(gq, l + 1, (l + 1) & 1);
                else if (has_next) issue_weights(g_next, l_next, 0);
            }
            mfmas(l & 1);
            phase_end_m();
            // ---------------- V: loads, E(gq, l), T(next step) ----------------
            if (kLate && l == 0 && gi > 0) store_out(co - 32, dc);
            float touched[4] = {0.f, 0.f, 0.f, 0.f};
            if (l == NL - 2) {
                load_old(co, dc);
                if (has_next) touch_x(g_next * 16 + li, touched);
            }
            epilogue(gq, l, slot + l, read_bn(gq, l));
            __builtin_amdgcn_sched_barrier(0);          // keep the transform's 24 patch reads (48 registers) behind the epilogue
            if (l + 1 < NL) {
                transform(l + 1);
            } else {
                // the group is complete: park old + total in dc (stored at the start of the next M phase), next group's x and transform
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) dc[r][hh][k] += total[r][hh][k];
                        total[r][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                if (has_next) {
                    load_x(g_next * 16 + li, xc);
                    transform(l_next);
                }
            }
            asm volatile("" ::"v"(touched[0]), "v"(touched[1]), "v"(touched[2]), "v"(touched[3]));
            if constexpr (kLate) {
                if (l + 2 < NL) { issue_weights(gq, l + 2, l & 1); phase_end_v3(); }
                else if (has_next && (l + 2 - NL < NL / 2 || gi + 1 < nfull)) { issue_weights(g_next, l_next + l + 2 - NL, l & 1); phase_end_v3(); }
                else phase_end_v();
            } else {
                phase_end_v();
            }
        }
        slot += NL;
    }
    if (odd) {
        // this worker's NL / 2 layers of the last group; its sum stays in `total` for the hand-over below
        const int gq = ngroups - 1;
        const int co = gq * 16 + li;
#pragma unroll
        for (int j = 0; j < NL / 2; ++j) {
            const int l = split_l0 + j;
            constexpr bool kLate = (OPT & 16) != 0;
            if (!kLate && j == 0 && nfull > 0) store_out((2 * (nfull - 1) + wk) * 16 + li, dc);
            if (!kLate && j + 1 < NL / 2) issue_weights(gq, l + 1, (j + 1) & 1);
            mfmas(j & 1);
            phase_end_m();
            if (kLate && j == 0 && nfull > 0) store_out((2 * (nfull - 1) + wk) * 16 + li, dc);
            if (j == 0 && wk == 0) load_old(co, dc);
            epilogue(gq, l, slot + j, read_bn(gq, l));
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < NL / 2) transform(l + 1);
            phase_end_v();
        }
        slot += NL / 2;
    }
    if (wk == 0) __builtin_amdgcn_s_barrier();          // the skew

    // ---- tail: the last parked result or the split group's hand-over, then the BN-backward sums ----
    if (!odd) {
        store_out((2 * (nfull - 1) + wk) * 16 + li, dc);
    } else {
        // worker 1's partial sum of the last group -> LDS (the U buffers are idle now) -> worker 0 adds and stores
        float* xch = s_u;
        if (wk == 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) *reinterpret_cast<f32x4*>(xch + ((2 * r + hh) * 256 + th) * 4) = total[r][hh];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x4 other = *reinterpret_cast<const f32x4*>(xch + ((2 * r + hh) * 256 + th) * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) dc[r][hh][k] += total[r][hh][k] + other[k];
                }
            store_out((ngroups - 1) * 16 + li, dc);
        }
    }
    if constexpr ((EXP & 4) == 0) {
        // one fp64 atomic per (worker, step, channel, sum): the 4 waves' partials added in a fixed order
        for (int i = tid; i < 2 * nsteps * 32; i += G::kThreads) {
            const int w = i / (nsteps * 32), rem = i - w * nsteps * 32;
            const int s = rem >> 5, j2 = rem & 31;
            const float* red = s_red + (w * G::kMaxSteps + s) * G::kRedStep + j2;
            const double t = static_cast<double>(red[0]) + static_cast<double>(red[32]) + static_cast<double>(red[64]) + static_cast<double>(red[96]);
            const int gi = s / NL;
            const int gq = gi < nfull ? 2 * gi + w : ngroups - 1;
            const int l = gi < nfull ? s - gi * NL : s - nfull * NL + w * (NL / 2);
            atomicAdd(p.scratch[l] + bn_slot_offset(p.slot_stride) + grp_off / 2 + 2 * (gq * 16) + j2, t);
        }
    }
}

inline bool dgrad_wino3_ok(const DgradBlockParams& p) {
    return dgrad_wino_ok(p) && p.count <= DgradWino3Geom<4>::kMaxCount;
}

// u[l]: transformed weights of layer l of the block in layout 1 (group-major slices of kWinoDgradSlice floats)
template <int NL, int EXP = 0, int OPT = 0>
inline int launch_dgrad_wino3(DgradBlockParams p, const float* const (&u)[4], hipStream_t stream) {
    using G = DgradWino3Geom<NL>;
    p.tiles_x = p.w / G::kTileX;
    const int tiles_y = p.h / G::kTileY;
    ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_wino3_kernel<NL, EXP, OPT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(G::kBytes)));
    dgrad_wino3_kernel<NL, EXP, OPT><<<dim3(p.tiles_x * tiles_y, 1, p.n), G::kThreads, G::kBytes, stream>>>(p, u[0], u[1], u[2], u[3]);
    ENDO_LAUNCH_CHECK();
    return 0;
}

}  // namespace endo
