cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4m/smoke.txt 2>&1; echo "smoke rc $?" >> gpurun_out/r4m/smoke.txt
tail -5 gpurun_out/r4m/smoke.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "kernel_forms" 2>&1 | tail -3
