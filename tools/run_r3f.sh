mkdir -p gpurun_out/r3f
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "fused_path or gap_scaled or warp_consistency or benchmark_batch or per_model or wgrad_overlap" > gpurun_out/r3f/pytest_new.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3f/pytest_new.txt
grep -v "^$" gpurun_out/r3f/pytest_new.txt | tail -40
python bench.py --steps 20 --warmup 3 --keep-graph-dot gpurun_out/r3f/step.dot > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err
tail -1 gpurun_out/r3f/bench.json | cut -c1-1500
head -c 3000 gpurun_out/r3f/step.dot
