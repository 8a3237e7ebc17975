#!/bin/bash
# stand-alone F(3x3,4x4) weight gradient before / after, then the in-job A/B and the weight-gradient tests
R=$PWD; O=$R/gpurun_out/r6q; mkdir -p $O
for b in x3_bench_old x3_bench; do
  echo "== $b" >> $O/x3.txt
  for a in "48 16 256 320" "84 16 256 320" "180 16 256 320" "228 16 256 320" "144 16 128 160" "228 16 128 160" "156 16 64 80" "264 16 64 80"; do
    $R/tools/bin/$b $a 2>&1 | grep "dense-layer\|F(3x3" | cut -c1-200 >> $O/x3.txt; done
done
cat $O/x3.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "wgrad or weight or fixture or transparent or training_step" 2>&1 | tail -5 > $O/t.txt; cat $O/t.txt
bash tools/ab.sh 3 $O/ab.txt "" base2 main; cat $O/ab.txt
