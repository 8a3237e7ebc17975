// The bf16-storage family compiled for IEEE half storage (BASELINE configs[4]: "mixed fp16 storage / fp32 accum"): the same sources with
// another element type and a power-of-two gradient scale (bf16_conv_kernels.h, "ELEMENT TYPE").  Exports endo_net16h_*.
#define ENDO16_HALF 1
#include "net16.hip"
