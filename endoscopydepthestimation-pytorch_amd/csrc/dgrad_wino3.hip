// The phase-skewed Winograd data-gradient kernel (dgrad_wino3_kernels.h) in a translation unit of its own: it is built with
// -fno-slp-vectorize (__graft_entry__.py).  The kernel lives at the edge of the 256-register budget of two waves per SIMD
// (48 A operands + 64 accumulators + 48 x / gradient / sum registers); clang's SLP vectoriser packs its scalar transform
// arithmetic into v_pk_add_f32 pairs that need v_mov shuffles and 25 more registers -- spills inside the phase loop, whose
// reloads wait on vmcnt(0) behind the weight DMA (measured in tools/wino_bench: 1.49 ms with SLP, 1.29 ms without, C0 = 144).
// The other kernels of net.hip keep the default (their packed transforms measured faster with it).
#include "dgrad_wino3_kernels.h"
#include "dgrad_wino3p_kernels.h"

namespace endo {

int run_dgrad_wino3_nl4(const DgradBlockParams& p, const float* const* u, hipStream_t stream) {
    const float* const uu[4] = {u[0], u[1], u[2], u[3]};
    return launch_dgrad_wino3<4>(p, uu, stream);
}

// the persistent-block form (dgrad_wino3p_kernels.h); fw_parts != nullptr: also the final convolution's weight-gradient partials
// (blocks_used x p.count doubles) of the range's channels -- the launch must carry the virtual final gradient (p.vg)
bool dgrad_wino3p_applies(const DgradBlockParams& p) { return dgrad_wino3p_ok(p); }

int run_dgrad_wino3p_nl4(const DgradBlockParams& p, const float* const* u, int blocks, double* fw_parts, int* blocks_used, hipStream_t stream) {
    const float* const uu[4] = {u[0], u[1], u[2], u[3]};
    if (fw_parts) return p.vg ? launch_dgrad_wino3p<4, true>(p, uu, blocks, fw_parts, blocks_used, stream) : ENDO_E_BADARG;
    return launch_dgrad_wino3p<4, false>(p, uu, blocks, nullptr, blocks_used, stream);
}

}  // namespace endo
