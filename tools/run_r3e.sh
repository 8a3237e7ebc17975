mkdir -p gpurun_out/r3e
python -m pytest tests -m gpu -x -q > gpurun_out/r3e/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r3e/pytest.txt
tail -15 gpurun_out/r3e/pytest.txt
tools/ab_options.sh 2 "wino3:" "wino8:1=2" > gpurun_out/r3e/ab_wino3.txt 2>&1
cat gpurun_out/r3e/ab_wino3.txt
