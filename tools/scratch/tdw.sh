#!/bin/bash
O=gpurun_out/r6r; mkdir -p $O
for a in "96 16 256 320" "144 16 128 160" "192 16 64 80" "240 16 32 40" "96 2 64 96" "100 3 32 40"; do timeout 120 tools/bin/tdw_bench $a; done > $O/tdw.txt 2>&1
cat $O/tdw.txt
