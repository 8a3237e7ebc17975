#!/bin/bash
# round 3, GPU run o: bf16 conv block-shape / prefetch-depth / weight-placement variants over 32-channel blocks
mkdir -p gpurun_out/r3o
for blk in 32; do
for cin in 180 48; do timeout 120 tools/bin/bf16_conv_variants $cin 16 256 320 $blk; done
timeout 120 tools/bin/bf16_conv_variants 224 16 128 160 $blk
done > gpurun_out/r3o/variants.txt 2>&1
cat gpurun_out/r3o/variants.txt
