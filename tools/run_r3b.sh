mkdir -p gpurun_out/r3b
for v in noslp slp; do for c in 144 48; do echo "== build $v"; timeout 300 tools/bin/wino_bench_$v $c 16 256 320 1; done; done > gpurun_out/r3b/wino_bench.txt 2>&1
cat gpurun_out/r3b/wino_bench.txt
