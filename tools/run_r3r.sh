#!/bin/bash
# round 3, GPU run r: bf16-storage tests + kernel trace of the bf16 forward + backward
mkdir -p gpurun_out/r3r
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -m gpu -s > gpurun_out/r3r/tests.txt 2>&1
grep -E "bf16-storage backward|passed|failed|network forward|forward of" gpurun_out/r3r/tests.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $R/tools/bf16_step_workload.py > $R/gpurun_out/r3r/prof.log 2>&1
cd $R
python3 tools/summarize_rocprof.py /tmp/kt/kt_results.db gpurun_out/r3r/kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python3 tools/bf16_step_workload.py (7 iterations)" > /dev/null 2>&1
python3 tools/summarize_rocprof.py --by-grid /tmp/kt/kt_results.db gpurun_out/r3r/kernel_stats_by_grid.txt > /dev/null 2>&1
head -40 gpurun_out/r3r/kernel_stats.txt | cut -c1-190
