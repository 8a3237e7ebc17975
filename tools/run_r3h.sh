mkdir -p gpurun_out/r3h
tools/ab_bench.sh 3 base prep ns768 ns1024 > gpurun_out/r3h/ab.txt 2>&1
cat gpurun_out/r3h/ab.txt
