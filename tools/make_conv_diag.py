#!/usr/bin/env python3
"""Development aid: a copy of csrc/bf16_conv_kernels.h with more diagnostic masks in the kEpiDgradBn epilogue, for
tools/bf16_dgrad_variants.hip (the product header stays as it is: its sha256 ties the committed profiles to the sources).
EXP bits added: 8 = no epilogue loads (forward values / old gradients), 16 = no gradient stores, 32 = no BatchNorm-backward sums,
64 = round to nearest instead of stochastic rounding.   usage: tools/make_conv_diag.py  ->  tools/bin/bf16_conv_diag_kernels.h"""
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(root, "endoscopydepthestimation-pytorch_amd", "csrc", "bf16_conv_kernels.h")).read()


def rep(a, b):
    global src
    assert src.count(a) == 1, (a, src.count(a))
    src = src.replace(a, b)


rep('''                        if (co < p.cout && offs[r][hh] >= 0) {
                            const int64_t o = out_ptr(offs[r][hh], co) - out_n;''',
    '''                        if ((EXP & 8) == 0 && co < p.cout && offs[r][hh] >= 0) {
                            const int64_t o = out_ptr(offs[r][hh], co) - out_n;''')
rep('''                        *reinterpret_cast<u32x2_t*>(dst) = u32x2_t{pack_s16x2_sr(nw[0], nw[1], key), pack_s16x2_sr(nw[2], nw[3], key + 2)};''',
    '''                        const u32x2_t st = (EXP & 64) ? u32x2_t{pack_s16x2(nw[0], nw[1]), pack_s16x2(nw[2], nw[3])}
                                                      : u32x2_t{pack_s16x2_sr(nw[0], nw[1], key), pack_s16x2_sr(nw[2], nw[3], key + 2)};
                        if ((EXP & 16) == 0 || st[0] == 0x12345678u) *reinterpret_cast<u32x2_t*>(dst) = st;''')
rep('''    if (p.out_sums) {
        // reduce over the 16 pixels of a lane group''', '''    if ((EXP & 32) == 0 && p.out_sums) {
        // reduce over the 16 pixels of a lane group''')
rep('''                if (off[i] >= 0) {
                    if (cpart < p.cin) { const u32x2_t h0''', '''                if ((EXP & 2) == 0 && off[i] >= 0) {
                    if (cpart < p.cin) { const u32x2_t h0''')
rep('#include "common.h"', '#include "../../endoscopydepthestimation-pytorch_amd/csrc/common.h"')
os.makedirs(os.path.join(root, "tools", "bin"), exist_ok=True)
open(os.path.join(root, "tools", "bin", "bf16_conv_diag_kernels.h"), "w").write(src)
print("wrote tools/bin/bf16_conv_diag_kernels.h")
