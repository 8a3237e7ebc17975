mkdir -p gpurun_out/r3a
( for c in 48 144; do timeout 300 tools/bin/wino_bench $c 16 256 320 0; done; for c in 96 192; do timeout 300 tools/bin/wino_bench $c 16 128 160 1; done ) > gpurun_out/r3a/wino_bench.txt 2>&1
tools/ab_bench.sh 2 main noslp > gpurun_out/r3a/ab_noslp.txt 2>&1
tail -50 gpurun_out/r3a/wino_bench.txt; cat gpurun_out/r3a/ab_noslp.txt
