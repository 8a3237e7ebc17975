mkdir -p gpurun_out/r3c
for c in 144 48; do timeout 300 tools/bin/wino_bench_noslp $c 16 256 320 1; done > gpurun_out/r3c/wino_bench.txt 2>&1
cat gpurun_out/r3c/wino_bench.txt | cut -c1-110
