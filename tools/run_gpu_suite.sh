cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite_r04d
timeout 1700 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/suite_r04d/pytest_tail.txt
cat gpurun_out/suite_r04d/pytest_tail.txt
