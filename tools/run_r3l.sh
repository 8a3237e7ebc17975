mkdir -p gpurun_out/r3l
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "two_streams or per_model" > gpurun_out/r3l/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3l/pytest.txt
tail -15 gpurun_out/r3l/pytest.txt
tools/ab_options.sh 3 "grouped:" "2streams:6=2" > gpurun_out/r3l/ab.txt 2>&1
cat gpurun_out/r3l/ab.txt
