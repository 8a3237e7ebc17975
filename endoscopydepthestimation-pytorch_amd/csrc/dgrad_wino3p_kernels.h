// Fused data gradient of a dense block's base channels, Winograd F(2x2, 3x3), phase-skewed -- as PERSISTENT blocks (round 6).
//
// Same arithmetic, work split and phase structure as dgrad_wino3_kernel (dgrad_wino3_kernels.h: read its header first).  That kernel
// keeps ONE 8-wave block per CU (150 KB of LDS), so whatever a block does before its first and after its last MFMA is exposed:
// tools/wino_bench's diagnostic builds priced the 66 KB dY tile load at 81 us of a level-0 launch (20 rounds of blocks per CU), the
// 1 152 fp64 atomics per block at 22 us, and a build without any global access ran 1040 us per-tile against 860 us persistent
// (C0 = 144).  Here a block stays on its CU and walks a contiguous run of tiles of ONE group of the batch:
//   * the BN table is built once per block;
//   * the dY tile of the NEXT tile is refilled layer by layer while the current tile's last steps run: the 12 maps of layer l are
//     dead as soon as both workers have transformed them for their last group (the last whole group of each worker, or -- odd group
//     count -- the split group), and worker 1's waves issue the DMA of the next tile's layer l at the start of the M phase that
//     follows (an M phase ends without a vmcnt wait: the DMA has that phase and the next V phase to land, and the V phase's closing
//     wait covers it; issued from a V phase -- first build -- the HBM latency sat in front of that phase's closing vmcnt(0));
//   * the first weight slice of the next tile goes out in the last M phase (buffer 0 is idle there: the last step of a tile always
//     runs on buffer 1);
//   * loads that return to registers trail their V phase: the old gradient of a group and the first touch of the next group's x are
//     the LAST vector-memory instructions of the V phase two steps before they are needed, x of the next group those of the group's
//     last V phase, and the phase closes with a COUNTED wait (vmcnt(8) / vmcnt(4): loads retire in order, so everything older -- the
//     weight slice, refills -- has landed) that leaves them in flight across the barrier and the following M phase.  In
//     dgrad_wino3_kernel they were issued at the start of a V phase and drained by its vmcnt(0): an L2 miss of 2 us in front of a
//     1.2 us phase;
//   * the BN-backward sums stay in LDS over the run (fp32 per (worker, step, wave, channel), owned slots: read-add-write, no atomics;
//     an fp64 LDS table fed by ds_add_f64 measured 8 % of the kernel) and leave every kFlushTiles tiles and at the end: the four
//     waves' slots added in fp64 in a fixed order, one fp64 atomic per (worker, step, channel, sum);
//   * FW (the last up block with the virtual final gradient, DgradBlockParams::vg): the final 1x1 convolution's weight gradient
//     dW[c] = sum over pixels of g * x[c] (reference models.py:167, 186) is formed here for the block's base channels -- the lane
//     holds x of its 16 pixels and loads the same 16 values of g for the virtual old gradient anyway: 16 FMAs per group, kept like the
//     BN sums, one fp64 partial per (block, channel) in `fw_parts` (block-private: plain read-add-write), added up in a fixed order
//     by final_w_reduce_kernel (net.hip).  final_bwd_weight_kernel then reads 48 instead of 192 planes.
// Every global access is buffer descriptor + ONE per-lane byte offset that never changes + a wave-uniform byte offset: nothing
// tile-dependent lives in vector registers (the pointer forms of dgrad_wino3_kernel, carried through this kernel's loop over tiles,
// spilled 30 registers into the phase loop).
// LDS: dY tile 66 KB + U 48 KB + BN table 12 KB + sums 24 KB + final-weight sums 3 KB = 153 KB (the split group's hand-over uses the
// idle weight buffers 1 of both workers).
#pragma once

#include <type_traits>

#include "dgrad_wino3_kernels.h"

namespace endo {

template <int NL>
struct DgradWino3PGeom {
    static_assert(NL == 4, "the trailing-load schedule is written for four layers");
    using B = DgradWino3Geom<NL>;
    static constexpr int kThreads = 512;
    static constexpr int kTileX = B::kTileX, kTileY = B::kTileY, kU = B::kU, kRedStep = B::kRedStep;
    // The dY maps arrive by 16-BYTE LDS-DMA: the tile starts 4 pixels left of the output tile, its rows are 10 aligned float4 units and a
    // map's 100 units are contiguous in LDS -- 2 DMA instructions per map instead of 6.  (Dword LDS-DMA runs at one 256-byte
    // instruction per ~40-55 cycles of a CU, measured with this kernel's barrier time stamps: the 288 instructions of a tile were 16 000
    // of its 145 000 cycles at C0 = 144 and 17 000 of 59 000 at C0 = 48, however early they were issued.)
    static constexpr int kLeft = 4, kCols = kTileX + 2 * kLeft, kRows = kTileY + 2;
    static constexpr int kPlane = kRows * kCols;                        // 400
    static constexpr int kUnits = kPlane / 4;                           // 100 float4 units per map: lanes 0..63, then 0..35
    static constexpr int kCS = 416;                                     // == 32 (mod 64) dwords: the patch reads of the 4 maps of a quad hit disjoint banks
    // A lane's patch starts at window column kLeft - 1 + 2 li = 3 + 2 li: every map sits ONE dword into its slot, so that the patch's
    // column pairs are 8-byte aligned (ds_read2_b64 as in dgrad_wino3_kernel; at odd dword offsets the 24 reads of a transform became
    // 24 ds_read2_b32 and every interval of the kernel 350 cycles longer).  The 16-byte DMA writes to a 4-byte aligned LDS address.
    static constexpr int kMapShift = 1;
    static_assert(kCS >= kPlane + kMapShift && kCS % 64 == 32, "dY map stride");
    static constexpr int kMaxCount = 144;                               // base channels (9 groups: the level-0 up block); wider blocks keep dgrad_wino3_kernel
    static constexpr int kMaxSteps = (kMaxCount / 32) * NL + NL / 2;    // per worker
    static constexpr int kG = NL * 12 * kCS;
    static constexpr int kMaxGroupSlots = kMaxCount / 32 + 1;           // per worker: its whole groups + the split group
    static constexpr int kFw = 2 * kMaxGroupSlots * 4 * 16;             // [worker][group slot][wave][16]
    static constexpr int kFw64 = 2 * kMaxCount;                          // FW: the block's running sums, one double per channel (owned by thread = channel)
    static constexpr int kTouch = 256;                                  // dummy target of the x touches (4 dword DMAs)
    static constexpr int kDbg = 2 * 64 * 2;                             // EXP & 16: barrier time stamps of one tile, [worker][64] x 8 bytes
    static constexpr int kFloats = kG + 4 * kU + NL * kMaxCount * 4 + 2 * kMaxSteps * kRedStep + kFw + kDbg + kTouch + kFw64;
    static constexpr size_t kBytes = sizeof(float) * kFloats;
    static constexpr int kFlushTiles = 4;                               // the final-conv weight sums leave the block every this many tiles (fp32 in between)
    static_assert(kBytes <= 160 * 1024, "one block per CU");
    static_assert(2 * 256 * 4 <= kU, "half of the split group's hand-over fits one weight buffer");
};

// p as for dgrad_wino3_kernel.  tiles_xy = tiles per sample, gn = samples per group, bpg = blocks per group (gridDim.x = bpg * groups).
// EXP: diagnostic bit mask for tools/wino_bench (0 in the library; timing only): 1 = no x / gradient loads, 2 = no stores,
// 4 = no BN-sum bookkeeping, 8 = no dY tile loads, 16 = block 0 records the clock behind every barrier of its third tile (fw_parts:
// 128 x uint64, [worker][barrier]) -- tools/wino_bench prints the phase lengths
template <int NL, bool FW = false, int EXP = 0>
__global__ void __launch_bounds__(512, 2) dgrad_wino3p_kernel(const DgradBlockParams p, const float* __restrict__ u0, const float* __restrict__ u1,
                                                              const float* __restrict__ u2, const float* __restrict__ u3, int tiles_xy, int gn, int bpg,
                                                              double* __restrict__ fw_parts) {
    using G = DgradWino3PGeom<NL>;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int grp = blockIdx.x / bpg;
    const int r0 = blockIdx.x - grp * bpg;
    const int64_t grp_off = grp * p.gs;
    const int t_total = tiles_xy * gn;
    // The blocks of an XCD (bpg / 8 per group) share one contiguous range of the group's tiles and walk it INTERLEAVED: block idx takes tiles
    // T0 + idx, T0 + idx + bpg / 8, ... -- at any time the XCD's blocks work on neighbouring tiles, whose haloed gradient windows (10 x 40 of
    // 8 x 32 pixels, 48 maps) then meet in the XCD's L2.  With a contiguous run per block a window's lines were fetched again by the block's
    // next tile ~60 us later, after 16 MB of other traffic: 1.8 GB of HBM fetch per level-0 launch for 1.1 GB algorithmic (r06_b PMC passes).
    int t_begin, t_end, t_step;
    if ((bpg & 7) == 0) {
        const int q = bpg >> 3, xcd = r0 & 7, idx = r0 >> 3;
        t_begin = static_cast<int>(static_cast<int64_t>(xcd * q) * t_total / bpg) + idx;
        t_end = static_cast<int>(static_cast<int64_t>((xcd + 1) * q) * t_total / bpg);
        t_step = q;
    } else {
        t_begin = static_cast<int>(static_cast<int64_t>(r0) * t_total / bpg);
        t_end = static_cast<int>(static_cast<int64_t>(r0 + 1) * t_total / bpg);
        t_step = 1;
    }

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_g = smem;                               // [NL*12][kCS]
    float* s_u = s_g + G::kG;                        // [worker][2][c 12][a 4][j 16][4]
    float* s_bn = s_u + 4 * G::kU;                   // [NL][count][scale, beta, mean, rstd]
    float* s_red = s_bn + NL * G::kMaxCount * 4;     // [worker][step][4 waves][16][2]
    float* s_fw = s_red + 2 * G::kMaxSteps * G::kRedStep;          // [worker][group slot][4 waves][16]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 2, w4 = wave & 3;         // worker; tile row of the wave
    const int th = tid & 255;
    const int li = lane & 15;
    const int lk = lane >> 4;
    const int ngroups = p.count / 16;
    const int nfull = ngroups >> 1;                  // whole groups per worker
    const bool odd = (ngroups & 1) != 0;
    const int nsteps = nfull * NL + (odd ? NL / 2 : 0);
    const float* const u_layer[4] = {u0, u1, u2, u3};
    const int split_l0 = wk * (NL / 2);              // first layer of this worker's part of a split group
    const int g_first = wk;                          // p.count >= 32: every worker has at least one whole group
    unsigned long long* s_dbg = reinterpret_cast<unsigned long long*>(s_fw + G::kFw);
    float* s_touch = s_fw + G::kFw + G::kDbg;
    double* s_fw64 = reinterpret_cast<double*>(s_touch + G::kTouch);
    int dbg_i = 0;
    bool dbg_on = false;
    auto stamp = [&]() {
        if constexpr ((EXP & 16) != 0) {
            if (dbg_on && w4 == 0 && dbg_i < 64) { if (lane == 0) s_dbg[wk * 64 + dbg_i] = __builtin_readcyclecounter(); ++dbg_i; }
        }
    };

    // ---- once per block: BN table, zeroed sums ----
    for (int i = tid; i < NL * p.count; i += G::kThreads) {
        const int l = i / p.count, ch = i - l * p.count;
        const float mean = p.saved[l][grp_off + 2 * ch], rstd = p.saved[l][grp_off + 2 * ch + 1];
        *reinterpret_cast<f32x4*>(s_bn + 4 * i) = f32x4{p.gamma[l][ch] * rstd, p.beta[l][ch], mean, rstd};
    }
    for (int i = tid; i < 2 * G::kMaxSteps * G::kRedStep + G::kFw; i += G::kThreads) s_red[i] = 0.f;
    if constexpr (FW) {
        for (int i = tid; i < p.count; i += G::kThreads) { fw_parts[static_cast<int64_t>(blockIdx.x) * p.count + i] = 0.0; s_fw64[i] = 0.0; }
    }
    if (t_begin >= t_end) return;          // (never with bpg <= tiles per group; block-uniform)

    auto rsrc = [](const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7ffffffc, 0x00020000); };
    auto ld4 = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); };
    auto ld1 = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); };
    auto st4 = [](const f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0); };

    // ---- the dY maps of a tile: chunk = 64 pixels of the haloed 10 x 34 plane, one dword DMA per (map, chunk) ----
    auto tile_origin = [&](int t, int& n, int& x0, int& y0) {
        n = t / tiles_xy;
        const int tile = t - n * tiles_xy;
        const int ty = tile / p.tiles_x;
        x0 = (tile - ty * p.tiles_x) * G::kTileX;
        y0 = ty * G::kTileY;
    };
    // The lane's two DMA units of a map of the tile at (x0, y0): unit u = lane (+ 64) = (row u / 10, float4 column u % 10) of the 10 x 40
    // window that starts at (y0 - 1, x0 - 4); byte offset inside a map, or -- outside the image, past unit 99 -- an offset past the
    // descriptor's range (the DMA then writes zeros)
    auto map_units = [&](int x0, int y0, unsigned (&vo)[2]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int u = lane + 64 * k;
            const int ry = u / (G::kCols / 4), rx = (u - ry * (G::kCols / 4)) * 4;
            const int gy = y0 - 1 + ry, gx = x0 - G::kLeft + rx;
            vo[k] = (u < G::kUnits && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w) ? 4u * static_cast<unsigned>(gy * p.g_w + gx) : 0x80000000u;
        }
    };
    auto map_rsrc = [&](const float* g_n) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g_n), 0, NL * 12 * p.g_cs * 4, 0x00020000); };
    // half k of map c (all 64 lanes issue: lanes past unit 99 write zeros into the map's padding / the next map's first bytes are never
    // reached: 64 + 36 units, the second instruction is masked to its 36 lanes)
    auto dma_map_half = [&](__amdgpu_buffer_rsrc_t gr, const unsigned (&vo)[2], int c, int k) {
        if constexpr ((EXP & 8) != 0) return;
        if (k == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lptr_t)(s_g + c * G::kCS + G::kMapShift), 16, vo[0], 4u * static_cast<unsigned>(c * p.g_cs), 0, 0);
        else if (lane < G::kUnits - 64) __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lptr_t)(s_g + c * G::kCS + G::kMapShift + 256), 16, vo[1], 4u * static_cast<unsigned>(c * p.g_cs), 0, 0);
    };
    // this worker's U slice of (group, layer) into weight buffer `buf`: one contiguous 12 KB run, 3 float4 units per thread
    auto issue_weights = [&](int gq, int l, int buf) {
        const __amdgpu_buffer_rsrc_t ur = rsrc(u_layer[l]);
        float* dst = s_u + (wk * 2 + buf) * G::kU + w4 * 256;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (lptr_t)(dst + k * 1024), 16, 16u * static_cast<unsigned>(th), 4u * static_cast<unsigned>(gq * G::kU + k * 1024), 0, 0);
    };

    const unsigned vlane_g = 4u * static_cast<unsigned>((2 * w4) * p.w + 8 * lk);          // the lane's first pixel inside a tile
    const unsigned vlane = vlane_g + 4u * static_cast<unsigned>(li) * static_cast<unsigned>(p.cs);          // ... of channel li of a group
    // (group, row r) of the tile whose first pixel lies tile_b bytes into a plane
    auto soff = [&](unsigned tb, int gq, int r) { return tb + 4u * static_cast<unsigned>(gq * 16 * p.cs + r * p.w); };

    // ---- per-tile state, all wave-uniform ----
    const float* x_n = nullptr;
    float* out_n = nullptr;
    const float* vg_n = nullptr;
    unsigned tile_b = 0;          // 4 * (y0 * w + x0)
    bool has_next_tile = false;
    int n_next = 0, x0_next = 0, y0_next = 0;

    auto load_x = [&](const float* base, unsigned tb, int gq, f32x4 (&dst)[2][2]) {
        const __amdgpu_buffer_rsrc_t xr = rsrc(base);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if constexpr ((EXP & 1) != 0) dst[r][hh] = f32x4{0.1f * lane, 0.2f, -0.3f, 0.4f};
                else dst[r][hh] = ld4(xr, vlane + 16u * hh, soff(tb, gq, r));
            }
    };
    // first touch of a later group's x (one dword per 16-byte unit): the real loads then come from L2.  As LDS-DMA into a dummy KB of
    // LDS: no destination registers, so nothing for the compiler to wait for (the register form needed a "use" two phases later, and
    // its wait -- counted over the shorter of two paths -- stalled on the refill DMAs issued in between)
    auto touch_x = [&](const float* base, unsigned tb, int gq) {
        if constexpr ((EXP & 1) != 0) return;
        const __amdgpu_buffer_rsrc_t xr = rsrc(base);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr_t)(s_touch + (2 * r + hh) * 64), 4, vlane + 16u * hh, soff(tb, gq, r), 0, 0);
    };
    // the old gradient of a group: the buffer's content, zeros (first writer), or -- p.vg -- the plane g of the virtual final gradient
    // (kept raw: the group's end multiplies it by the channel's final-conv weight and, FW, forms g . x).  Always 4 loads.
    float wf_cur = 0.f;          // p.vw[co] of the group whose old gradient is in flight
    auto load_wf = [&](int gq) { if (vg_n) wf_cur = ld1(rsrc(p.vw), 4u * static_cast<unsigned>(li), 64u * static_cast<unsigned>(gq)); };
    // (a group none of whose channels has a gradient yet -- acc_from -- issues no loads: all_fresh, block-uniform)
    auto all_fresh = [&](int gq) { return !vg_n && gq * 16 + 15 < p.acc_from; };
    auto load_old = [&](int gq, f32x4 (&dst)[2][2]) {
        const __amdgpu_buffer_rsrc_t orr = rsrc(vg_n ? static_cast<const void*>(vg_n) : static_cast<const void*>(out_n));
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if ((EXP & 1) != 0) dst[r][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
                else if (vg_n) dst[r][hh] = ld4(orr, vlane_g + 16u * hh, tile_b + 4u * static_cast<unsigned>(r * p.w));
                else dst[r][hh] = ld4(orr, vlane + 16u * hh, soff(tile_b, gq, r));
            }
    };
    auto zero_old = [&](f32x4 (&dst)[2][2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) dst[r][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_out = [&](int gq, const f32x4 (&src)[2][2]) {
        const __amdgpu_buffer_rsrc_t orr = rsrc(out_n);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if constexpr ((EXP & 2) != 0) asm volatile("" ::"v"(src[r][hh][0]), "v"(src[r][hh][1]), "v"(src[r][hh][2]), "v"(src[r][hh][3]));
                // (offset all in the vector register: behind a 16-byte store with a SCALAR offset register the compiler inserts no wait state before
                // a VALU write of the data registers -- td_dgrad_kernels.h lost the first dword of 0.1 % of such stores)
                else st4(src[r][hh], orr, vlane + 16u * hh + soff(tile_b, gq, r), 0u);
            }
    };

    f32x4 xc[2][2], dc[2][2], total[2][2];           // [row][column half]: 8 consecutive pixels of 2 rows
    f32x2 av[3][8];
    f32x4 acc[16];

    // ---- T: input transform of the lane's 4x4 patches of layer l's 12 maps (dgrad_wino3_kernel's, unchanged) ----
    auto load_patches = [&](int l) {
#pragma unroll
        for (int quad = 0; quad < 3; ++quad) {
            // rows 2 w4 .. 2 w4 + 3, columns x0 - 1 + 2 li .. + 3 = window columns 3 + 2 li ..
            const float* a_base = s_g + (l * 12 + quad * 4 + lk) * G::kCS + G::kMapShift + (2 * w4) * G::kCols + (G::kLeft - 1) + 2 * li;
#pragma unroll
            for (int row = 0; row < 4; ++row) {
                av[quad][2 * row] = *reinterpret_cast<const f32x2*>(a_base + row * G::kCols);
                av[quad][2 * row + 1] = *reinterpret_cast<const f32x2*>(a_base + row * G::kCols + 2);
            }
        }
    };
    auto transform_inplace = [&]() {
#pragma unroll
        for (int quad = 0; quad < 3; ++quad) {
            auto sub2 = [](const f32x2 a, const f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; };
            const f32x2 l0 = av[quad][0], h0 = av[quad][1], l1 = av[quad][2], h1 = av[quad][3];
            const f32x2 l2 = av[quad][4], h2 = av[quad][5], l3 = av[quad][6], h3 = av[quad][7];
            const f32x2 tl[4] = {sub2(l0, l2), l1 + l2, sub2(l2, l1), sub2(l1, l3)};
            const f32x2 th2[4] = {sub2(h0, h2), h1 + h2, sub2(h2, h1), sub2(h1, h3)};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x2 v01, v23;
                asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(v01) : "v"(tl[a]), "v"(th2[a]));
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v23) : "v"(th2[a]), "v"(tl[a]));
                av[quad][2 * a] = v01;
                av[quad][2 * a + 1] = v23;
            }
        }
    };
    auto transform = [&](int l) { load_patches(l); transform_inplace(); };

    // ---- M: 16 transform-domain GEMMs over the layer's 12 dY maps.  NJ > 0: the wave also issues NJ refill DMAs of the next tile -- both
    //         halves of the NJ / 2 maps from c0 on -- spread over its 48 MFMAs: a 16-byte LDS-DMA instruction costs the CU's address path
    //         ~70 cycles (time stamps: 48 of them issued by four waves in front of their MFMAs lengthened the interval by 3 500 cycles),
    //         so a burst in front of the MFMAs is on the block's critical path and more than ~one per 70 cycles and CU backs up ----
    auto mfmas = [&](int buf, auto nj_tag, __amdgpu_buffer_rsrc_t gr, const unsigned (&vo)[2], int c0) {
        constexpr int NJ = decltype(nj_tag)::value;
        static_assert(NJ == 0 || NJ == 6 || NJ == 8, "jobs per M phase");
        const float* ub = s_u + (wk * 2 + buf) * G::kU;
#pragma unroll
        for (int quad = 0; quad < 3; ++quad) {
            const float* b_base = ub + ((quad * 4 + lk) * 4 * 16 + li) * 4;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(b_base + a * 64);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (quad == 0) acc[4 * a + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[quad][2 * a + (i >> 1)][i & 1], b[i], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    else acc[4 * a + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[quad][2 * a + (i >> 1)][i & 1], b[i], acc[4 * a + i], 0, 0, 0);
                }
                if constexpr (NJ > 0) {
                    const int pair = quad * 4 + a;          // 12 (quad, a) pairs of 4 MFMAs: NJ = 6 -> behind every second, NJ = 8 -> behind two of three
                    const bool here = NJ == 6 ? (pair & 1) == 1 : (pair % 3) != 1;
                    if (here) {
                        const int job = NJ == 6 ? pair / 2 : pair - (pair + 1) / 3;
                        dma_map_half(gr, vo, c0 + (job >> 1), job & 1);
                    }
                }
            }
        }
    };
    using Tag0 = std::integral_constant<int, 0>;
    using Tag6 = std::integral_constant<int, 6>;
    using Tag8 = std::integral_constant<int, 8>;

    // sum over the wave's four 16-lane rows, in every lane: v_permlane16_swap / v_permlane32_swap (gfx950) exchange rows between two
    // registers in the VALU; __shfl_xor is a ds_bpermute round trip through the LDS crossbar (four dependent ones per step were ~300 of
    // an interval's ~3 300 cycles: the time stamps with and without the sums)
    // (inline assembly: through __builtin_amdgcn_permlane16_swap this compiler added the FIRST result to itself -- both results of a swap of
    // a value with itself are folded into one; the s_nop covers the VALU-write -> permlane-read wait states the compiler inserts itself)
    auto rows_sum = [](float v) {
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));          // (r0, r0, r2, r2), (r1, r1, r3, r3)
        const float t = a + b;
        float c = t, d = t;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));          // (t01 in every row), (t23 in every row)
        return c + d;
    };

    // ---- E: output transform, layer l's ReLU mask + BN backward, accumulated over the layers; the step's sums join the wave's slot ----
    auto read_bn = [&](int gq, int l) { return *reinterpret_cast<const f32x4*>(s_bn + 4 * (l * p.count + gq * 16 + li)); };
    auto epilogue = [&](int slot, const f32x4 bn) {
        const f32x2 sb = {bn[0], bn[1]}, mr = {bn[2], bn[3]};          // (scale, beta), (mean, rstd)
        const float rstd = bn[3];
        f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
        f32x2 minus1 = {-1.f, -1.f};
        asm("" : "+v"(minus1));
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x2 u0r[4], u1r[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x2 m0 = {acc[c][2 * hh], acc[c][2 * hh + 1]}, m1 = {acc[4 + c][2 * hh], acc[4 + c][2 * hh + 1]};
                const f32x2 m2 = {acc[8 + c][2 * hh], acc[8 + c][2 * hh + 1]}, m3 = {acc[12 + c][2 * hh], acc[12 + c][2 * hh + 1]};
                u0r[c] = m0 + m1 + m2;
                u1r[c] = __builtin_elementwise_fma(m2 + m3, minus1, m1);
            }
            const f32x2 d[2][2] = {{u0r[0] + u0r[1] + u0r[2], __builtin_elementwise_fma(u0r[2] + u0r[3], minus1, u0r[1])},
                                   {u1r[0] + u1r[1] + u1r[2], __builtin_elementwise_fma(u1r[2] + u1r[3], minus1, u1r[1])}};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int ee = 0; ee < 2; ++ee) {
                    const f32x2 x2 = {xc[r][hh][2 * ee], xc[r][hh][2 * ee + 1]};
                    f32x2 xcen, z;
                    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(xcen) : "v"(x2), "v"(mr));
                    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(z) : "v"(xcen), "v"(sb));
                    const f32x2 dz = {z[0] > 0.f ? d[r][0][ee] : 0.f, z[1] > 0.f ? d[r][1][ee] : 0.f};
                    s1v += dz;
                    s2v = __builtin_elementwise_fma(dz, xcen, s2v);
                    f32x2 t2 = {total[r][hh][2 * ee], total[r][hh][2 * ee + 1]};
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(t2) : "v"(dz), "v"(sb));
                    total[r][hh][2 * ee] = t2[0];
                    total[r][hh][2 * ee + 1] = t2[1];
                }
            }
        }
        float s1 = s1v[0] + s1v[1], s2 = s2v[0] + s2v[1];
        s2 *= rstd;
        s1 = rows_sum(s1);
        s2 = rows_sum(s2);
        if (lk == 0 && (EXP & 4) == 0) {
            f32x2* red = reinterpret_cast<f32x2*>(s_red + ((wk * G::kMaxSteps + slot) * 4 + w4) * 32 + 2 * li);
            *red = f32x2{s1, s2};
        }
    };
    // a group is complete: old + total -> dc (stored at the start of the next M phase); with the virtual final gradient dc holds the raw
    // plane g until here: g * w_final[co] is the old gradient and, FW, sum g * x the channel's final-conv weight gradient
    auto finish_group = [&](int gslot, int gq) {
        if (vg_n) {
            const float wf = wf_cur;
            float fw = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if constexpr (FW) fw = __builtin_fmaf(dc[r][hh][k], xc[r][hh][k], fw);
                        // the product rounded as final_bwd_data_kernel would have stored it (bit-identical to the materialised form): the empty
                        // asm keeps the compiler from contracting it into an FMA with the sum (__fmul_rn / __fadd_rn are plain operators in HIP)
                        float old_g = dc[r][hh][k] * wf;
                        asm("" : "+v"(old_g));
                        dc[r][hh][k] = old_g + total[r][hh][k];
                    }
            if constexpr (FW) {
                fw = rows_sum(fw);
                if (lk == 0) {
                    float* a = s_fw + ((wk * G::kMaxGroupSlots + gslot) * 4 + w4) * 16 + li;
                    *a += fw;
                }
            }
        } else {
            const bool fresh = gq * 16 + li < p.acc_from;          // (channels that have no gradient yet: whatever the load returned is dropped)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int k = 0; k < 4; ++k) dc[r][hh][k] = fresh ? total[r][hh][k] : dc[r][hh][k] + total[r][hh][k];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) total[r][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // A V phase closes with vmcnt(N) lgkmcnt(0) + the block barrier: N = the youngest vector-memory instructions it leaves in flight --
    // the register loads it has just issued for a later phase and, in front of them, the refill DMAs of the wave's last M phase (needed by
    // the NEXT tile: its opening wait covers them).  Loads retire in order: the weight slice -- issued ahead of the refills -- has landed.
    // The builtin, not inline assembly: the compiler's own wait-count bookkeeping sees it and places the waits of the loads left in flight
    // at their first use
    auto phase_end_v = [&](auto n_tag) {
        constexpr int N = decltype(n_tag)::value;
        static_assert(N >= 0 && N < 64, "vmcnt");
        __builtin_amdgcn_s_waitcnt((N & 15) | 0x0070 | ((N >> 4) << 14));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stamp();
    };
    // base = the register loads, nj = the refill DMAs of this wave's last M phase (0, 6, 8; wave-uniform, known where the phase ends)
    auto phase_end_vn = [&](auto base_tag, int nj) {
        constexpr int Bn = decltype(base_tag)::value;
        if (nj == 0) phase_end_v(std::integral_constant<int, Bn>{});
        else if (nj == 6) phase_end_v(std::integral_constant<int, Bn + 6>{});
        else phase_end_v(std::integral_constant<int, Bn + 8>{});
    };
    using V0 = std::integral_constant<int, 0>;
    using V4 = std::integral_constant<int, 4>;
    using V8 = std::integral_constant<int, 8>;
    auto phase_end_m = [&]() {
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0) only: DMA stays in flight across the barrier
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stamp();
    };

    // ---- first tile: everything from scratch ----
    {
        int n, x0, y0;
        tile_origin(t_begin, n, x0, y0);
        {   // all NL * 12 maps, 6 per wave
            unsigned vo[2];
            map_units(x0, y0, vo);
            const __amdgpu_buffer_rsrc_t gr = map_rsrc(p.g + grp_off + static_cast<int64_t>(n) * p.g_ns);
#pragma nounroll
            for (int c = wave * (NL * 12 / 8); c < (wave + 1) * (NL * 12 / 8); ++c) { dma_map_half(gr, vo, c, 0); dma_map_half(gr, vo, c, 1); }
        }
        issue_weights(g_first, 0, 0);
    }

    int since_flush = 0;
    for (int t = t_begin; t < t_end; t += t_step) {
        {
            int n, x0, y0;
            tile_origin(t, n, x0, y0);
            x_n = p.x + grp_off + static_cast<int64_t>(n) * p.ns;
            out_n = p.out + grp_off + static_cast<int64_t>(n) * p.ns;
            vg_n = p.vg ? p.vg + grp_off + static_cast<int64_t>(n) * p.cs : nullptr;
            tile_b = 4u * static_cast<unsigned>(y0 * p.w + x0);
            has_next_tile = t + t_step < t_end;
            if (has_next_tile) tile_origin(t + t_step, n_next, x0_next, y0_next);
        }
        // worker 1's four waves refill the next tile's maps from inside their M phases (mfmas): the lane's units and the descriptor
        unsigned vo_next[2] = {0x80000000u, 0x80000000u};
        if (has_next_tile && (wk == 1 || odd)) map_units(x0_next, y0_next, vo_next);
        const __amdgpu_buffer_rsrc_t gr_next = map_rsrc(p.g + grp_off + static_cast<int64_t>(has_next_tile ? n_next : 0) * p.g_ns);
        // x of the next tile's first group (has_next_tile), or -- keeping the phase's load count fixed -- of this tile's
        auto x_base_after = [&]() { return has_next_tile ? p.x + grp_off + static_cast<int64_t>(n_next) * p.ns : x_n; };
        auto tile_b_after = [&]() { return has_next_tile ? 4u * static_cast<unsigned>(y0_next * p.w + x0_next) : tile_b; };
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) { total[r][hh] = f32x4{0.f, 0.f, 0.f, 0.f}; dc[r][hh] = total[r][hh]; }
        // x of the first group: needed by E(0), behind T(0) and M(0) -- left in flight; everything older (the previous tile's stores and
        // atomics, the refills, the first weight slice) has landed behind this wait
        __builtin_amdgcn_sched_barrier(0);
        load_x(x_n, tile_b, g_first, xc);
        __builtin_amdgcn_s_waitcnt(0x0074);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();          // the tile's dY maps (first load or refills) and its first weight slice are in place (a raw barrier:
        __builtin_amdgcn_sched_barrier(0);     // __syncthreads()'s fence would wait for the loads just issued)
        if constexpr ((EXP & 16) != 0) { dbg_on = blockIdx.x == 0 && t == t_begin + 2 * t_step; dbg_i = 0; }
        stamp();
        if (wk == 1) { __builtin_amdgcn_s_barrier(); stamp(); }          // the skew
        transform(0);
        phase_end_v(V0{});

        int slot = 0;
        for (int gi = 0; gi < nfull; ++gi) {
            const int gq = 2 * gi + wk;
            const bool has_next = gi + 1 < nfull || odd;
            const int g_next_q = gi + 1 < nfull ? gq + 2 : ngroups - 1;
            const int l_next = gi + 1 < nfull ? 0 : split_l0;
            const bool last_whole = !odd && gi + 1 == nfull;          // this worker's last steps on the tile: the dY maps die layer by layer
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                // ---------------- M(gq, l): parked stores, next weight slice, [refill], 48 MFMAs ----------------
                if (l == 0 && gi > 0) store_out(gq - 2, dc);
                if (l + 1 < NL) issue_weights(gq, l + 1, (l + 1) & 1);
                else if (has_next) issue_weights(g_next_q, l_next, 0);
                else if (has_next_tile) issue_weights(g_first, 0, 0);
                const int nj = (last_whole && wk == 1 && has_next_tile) ? 6 : 0;          // even group count: layer l of the next tile, 3 maps per wave of worker 1
                if (nj) mfmas(l & 1, Tag6{}, gr_next, vo_next, l * 12 + 3 * w4);
                else mfmas(l & 1, Tag0{}, gr_next, vo_next, 0);
                phase_end_m();
                // ---------------- V: E(gq, l), T(next step), the loads that trail the phase ----------------
                if (l == 1) load_wf(gq);
                epilogue(slot + l, read_bn(gq, l));
                __builtin_amdgcn_sched_barrier(0);          // keep the transform's 24 patch reads (48 registers) behind the epilogue
                if (l + 1 < NL) {
                    transform(l + 1);
                } else {
                    finish_group(gi, gq);
                    if (has_next) transform(l_next);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (l == 1) {
                    const bool fresh_group = all_fresh(gq);
                    if (fresh_group) { zero_old(dc); __builtin_amdgcn_sched_barrier(0); }          // (before the loads: a register write of dc waits for whatever the compiler thinks may still be landing there)
                    if (has_next) touch_x(x_n, tile_b, g_next_q);
                    else touch_x(x_base_after(), tile_b_after(), g_first);
                    if (fresh_group) phase_end_vn(V4{}, nj);
                    else { load_old(gq, dc); phase_end_vn(V8{}, nj); }
                } else if (l == 3) {
                    if (has_next) { load_x(x_n, tile_b, g_next_q, xc); phase_end_vn(V4{}, nj); }
                    else phase_end_vn(V0{}, nj);
                } else {
                    phase_end_vn(V0{}, nj);
                }
            }
            slot += NL;
        }
        if (odd) {
            // this worker's NL / 2 layers of the last group; its sum stays in `total` for the hand-over below.  The dY maps die in the
            // order (0, 2), (1, 3): worker 0 transforms layers 0, 1 and worker 1 layers 2, 3 of this group, a phase apart
            const int gq = ngroups - 1;
#pragma unroll
            for (int j = 0; j < NL / 2; ++j) {
                const int l = split_l0 + j;
                if (j == 0) store_out(2 * (nfull - 1) + wk, dc);
                if (j + 1 < NL / 2) issue_weights(gq, l + 1, (j + 1) & 1);
                else if (has_next_tile) issue_weights(g_first, 0, 0);
                // Refills, odd group count: the maps die in the order (0, 2), (1, 3) over these two steps; the 96 DMAs of the next tile go
                // out in three M phases, 8 per wave: worker 1's M(j = 0) [layer 0, layer 2 maps 0-3], worker 0's M(j = 1) [layer 2 maps
                // 4-11, layer 1 maps 0-7], worker 1's M(j = 1) [layer 1 maps 8-11, layer 3] -- each wave 4 consecutive maps of one layer
                int nj = 0, c0 = 0;
                if (has_next_tile && (wk == 1 || j == 1)) {
                    nj = 8;
                    const int phase = wk == 0 ? 1 : 2 * j;          // A = 0, B = 1, C = 2
                    const int first = phase * 16 + 4 * w4;          // index into [L0 0-11, L2 0-3 | L2 4-11, L1 0-7 | L1 8-11, L3 0-11]
                    c0 = first < 12 ? first : first < 24 ? 24 + (first - 12) : first < 36 ? 12 + (first - 24) : 36 + (first - 36);
                }
                if (nj) mfmas(j & 1, Tag8{}, gr_next, vo_next, c0);
                else mfmas(j & 1, Tag0{}, gr_next, vo_next, 0);
                phase_end_m();
                if (j == 0 && wk == 0) load_wf(gq);
                epilogue(slot + j, read_bn(gq, l));
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < NL / 2) transform(l + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0) {
                    const bool with_old = wk == 0 && !all_fresh(gq);
                    if (wk == 0 && !with_old) { zero_old(dc); __builtin_amdgcn_sched_barrier(0); }
                    touch_x(x_base_after(), tile_b_after(), g_first);
                    if (with_old) { load_old(gq, dc); phase_end_vn(V8{}, nj); }
                    else phase_end_vn(V4{}, nj);
                } else {
                    phase_end_vn(V0{}, nj);
                }
            }
            slot += NL / 2;
        }
        if (wk == 0) { __builtin_amdgcn_s_barrier(); stamp(); }          // the skew

        // ---- tail of the tile: the last parked result or the split group's hand-over; the sums leave every kFlushTiles tiles ----
        if (!odd) {
            store_out(2 * (nfull - 1) + wk, dc);
        } else {
            // worker 1's partial sum of the last group -> the idle weight buffers 1 of both workers -> worker 0 adds and stores
            float* xa = s_u + 1 * G::kU;          // worker 0, buffer 1
            float* xb = s_u + 3 * G::kU;          // worker 1, buffer 1
            if (wk == 1) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    *reinterpret_cast<f32x4*>(xa + (hh * 256 + th) * 4) = total[0][hh];
                    *reinterpret_cast<f32x4*>(xb + (hh * 256 + th) * 4) = total[1][hh];
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the hand-over is written; a raw barrier (refill DMAs stay in flight)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (wk == 0) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x4 o0 = *reinterpret_cast<const f32x4*>(xa + (hh * 256 + th) * 4);
                    const f32x4 o1 = *reinterpret_cast<const f32x4*>(xb + (hh * 256 + th) * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { total[0][hh][k] += o0[k]; total[1][hh][k] += o1[k]; }
                }
                finish_group(nfull, ngroups - 1);
                store_out(ngroups - 1, dc);
            }
        }
        {
            // every step's epilogue lies behind a block barrier here (the next tile's opening barrier orders these reads before its
            // epilogues' stores).  One fp64 atomic per (worker, step, channel, sum): the 4 waves' partials added in a fixed order
            if constexpr ((EXP & 4) == 0) {
                for (int i = tid; i < 2 * nsteps * 32; i += G::kThreads) {
                    const int w = i / (nsteps * 32), rem = i - w * nsteps * 32;
                    const int s = rem >> 5, j2 = rem & 31;
                    float* red = s_red + (w * G::kMaxSteps + s) * G::kRedStep + j2;
                    const double v = static_cast<double>(red[0]) + static_cast<double>(red[32]) + static_cast<double>(red[64]) + static_cast<double>(red[96]);
                    const int gi = s / NL;
                    const int gq = gi < nfull ? 2 * gi + w : ngroups - 1;
                    const int l = gi < nfull ? s - gi * NL : s - nfull * NL + w * (NL / 2);
                    double* sc = l == 0 ? p.scratch[0] : l == 1 ? p.scratch[1] : l == 2 ? p.scratch[2] : p.scratch[3];
                    atomicAdd(sc + bn_slot_offset(p.slot_stride) + grp_off / 2 + 2 * (gq * 16) + j2, v);
                }
            }
            if (FW && (!has_next_tile || ++since_flush == G::kFlushTiles)) {
                since_flush = 0;
                for (int i = tid; i < p.count; i += G::kThreads) {
                    const int gq = i >> 4, j = i & 15;
                    const int w = (gq == ngroups - 1 && odd) ? 0 : (gq & 1);
                    const int gslot = (gq == ngroups - 1 && odd) ? nfull : (gq >> 1);
                    float* a = s_fw + (w * G::kMaxGroupSlots + gslot) * 64 + j;
                    const double v = static_cast<double>(a[0]) + static_cast<double>(a[16]) + static_cast<double>(a[32]) + static_cast<double>(a[48]);
                    a[0] = 0.f; a[16] = 0.f; a[32] = 0.f; a[48] = 0.f;
                    const double run = s_fw64[i] + v;          // (thread i owns channel i: no atomics; the block's partial leaves once, behind its last tile)
                    s_fw64[i] = run;
                    if (!has_next_tile) fw_parts[static_cast<int64_t>(blockIdx.x) * p.count + i] = run;
                }
            }
        }
    }
    if constexpr ((EXP & 16) != 0) {
        __syncthreads();
        if (blockIdx.x == 0 && tid < 128) reinterpret_cast<unsigned long long*>(fw_parts)[tid] = s_dbg[tid];
    }
}

// (the dY maps of a sample are addressed through one buffer descriptor: 32-bit byte range)
inline bool dgrad_wino3p_ok(const DgradBlockParams& p) { return dgrad_wino3_ok(p) && p.count <= DgradWino3PGeom<4>::kMaxCount && 48ll * p.g_cs * 4 < (1ll << 31) && static_cast<int64_t>(p.count) * p.cs * 4 < (1ll << 31); }

// blocks: persistent blocks of the launch (<= one per CU; rounded down to a multiple of 8 per group).  fw_parts (FW): blocks x p.count doubles.
template <int NL, bool FW = false, int EXP = 0>
inline int launch_dgrad_wino3p(DgradBlockParams p, const float* const (&u)[4], int blocks, double* fw_parts, int* blocks_used, hipStream_t stream) {
    using G = DgradWino3PGeom<NL>;
    if (!dgrad_wino3p_ok(p)) return ENDO_E_UNSUPPORTED;
    p.tiles_x = p.w / G::kTileX;
    const int tiles_xy = p.tiles_x * (p.h / G::kTileY);
    const int groups = p.group_n > 0 ? p.n / p.group_n : 1;
    const int gn = p.group_n > 0 ? p.group_n : p.n;
    int bpg = blocks / groups;
    if (bpg >= 8) bpg &= ~7;
    if (bpg > tiles_xy * gn) bpg = tiles_xy * gn;
    if (bpg < 1) bpg = 1;
    if (blocks_used) *blocks_used = bpg * groups;
    ENDO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_wino3p_kernel<NL, FW, EXP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(G::kBytes)));
    dgrad_wino3p_kernel<NL, FW, EXP><<<dim3(bpg * groups), G::kThreads, G::kBytes, stream>>>(p, u[0], u[1], u[2], u[3], tiles_xy, gn, bpg, fw_parts);
    ENDO_LAUNCH_CHECK();
    return 0;
}

}  // namespace endo
