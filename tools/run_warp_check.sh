cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "warp_consistency or test_losses or loss_head" > gpurun_out/r4f/pytest_warp.txt 2>&1; echo "rc $?" >> gpurun_out/r4f/pytest_warp.txt
tail -15 gpurun_out/r4f/pytest_warp.txt
