mkdir -p gpurun_out/r3k
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "network_forward_levels or kernel_forms or network_golden or full_size_pair_backward" > gpurun_out/r3k/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3k/pytest.txt
tail -5 gpurun_out/r3k/pytest.txt
tools/ab_bench.sh 3 base winof > gpurun_out/r3k/ab.txt 2>&1
cat gpurun_out/r3k/ab.txt
