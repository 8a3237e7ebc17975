// The phase-skewed Winograd data-gradient kernel (dgrad_wino3_kernels.h) in a translation unit of its own: it is built with
// -fno-slp-vectorize (__graft_entry__.py).  The kernel lives at the edge of the 256-register budget of two waves per SIMD
// (48 A operands + 64 accumulators + 48 x / gradient / sum registers); clang's SLP vectoriser packs its scalar transform
// arithmetic into v_pk_add_f32 pairs that need v_mov shuffles and 25 more registers -- spills inside the phase loop, whose
// reloads wait on vmcnt(0) behind the weight DMA (measured in tools/wino_bench: 1.49 ms with SLP, 1.29 ms without, C0 = 144).
// The other kernels of net.hip keep the default (their packed transforms measured faster with it).
#include "dgrad_wino3_kernels.h"

namespace endo {

int run_dgrad_wino3_nl4(const DgradBlockParams& p, const float* const* u, hipStream_t stream) {
    const float* const uu[4] = {u[0], u[1], u[2], u[3]};
    return launch_dgrad_wino3<4>(p, uu, stream);
}

}  // namespace endo
