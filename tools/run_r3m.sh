mkdir -p gpurun_out/r3m
python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -s > gpurun_out/r3m/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3m/pytest.txt
grep -v "^$" gpurun_out/r3m/pytest.txt | tail -30
