#!/bin/bash
O=gpurun_out/r6al; mkdir -p $O
for b in conv_bench conv_bench_new; do echo "== $b"; for a in "180 16 64 80" "228 16 32 40" "276 16 16 20"; do timeout 200 tools/bin/$b $a 2>&1 | grep "dense layer\|(library)\|KC16 2buf 16x8$\|td fwd KC8 Q3 16x8 (lib" | cut -c1-120; done; done > $O/cb.txt 2>&1
cat $O/cb.txt
bash tools/ab.sh 3 $O/ab.txt "" tud main; cat $O/ab.txt
timeout 1500 python -m pytest tests -q -m gpu -x -k "not storage and not bf16 and not fp16" 2>&1 | tail -3
