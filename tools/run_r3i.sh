mkdir -p gpurun_out/r3i
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3i/kt.err
python3 $GRAFT_REPO_ROOT/tools/stream_timeline.py /tmp/kt/kt_results.db > $GRAFT_REPO_ROOT/gpurun_out/r3i/timeline.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r3i/timeline.txt
