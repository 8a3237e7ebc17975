mkdir -p gpurun_out/r3g
python tests/diag/dispatch_count_probe.py > gpurun_out/r3g/dispatch_probe.txt 2>&1
tail -40 gpurun_out/r3g/dispatch_probe.txt
tools/ab_bench.sh 2 base nsred noslp > gpurun_out/r3g/ab_nsred.txt 2>&1
cat gpurun_out/r3g/ab_nsred.txt
cd /tmp && export TMPDIR=/tmp
for v in base noslp; do
  ENDO_HIP_LIB=$GRAFT_REPO_ROOT/tools/bin/libendo_hip_$v.so rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3g/kt_$v.err
  python3 $GRAFT_REPO_ROOT/tools/summarize_rocprof.py /tmp/kt_$v/kt_results.db $GRAFT_REPO_ROOT/gpurun_out/r3g/kernel_stats_$v.txt "variant $v"
done
