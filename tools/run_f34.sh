cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "kernel_forms and winograd4-1x64x128 or test_train_step_full_size_golden" 2>&1 | grep -v "   kept" | grep "Error\|assert\|worst\|over\|FAILED\|passed\|failed\|rel\|tol" | cut -c1-400 | head -40
