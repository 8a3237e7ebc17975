set -x
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
python3 $R/tools/summarize_rocprof.py /tmp/kt/kt_results.db $O/${TAG}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline" 7
python3 $R/tools/chain_trace.py /tmp/kt/kt_results.db 2 > $O/${TAG}_chain_trace.txt 2>&1
python3 $R/tools/stream_timeline.py /tmp/kt/kt_results.db > $O/${TAG}_stream_timeline.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/kts -o kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel-option 5=0 > /dev/null 2> $O/kts.err
python3 $R/tools/summarize_rocprof.py /tmp/kts/kt_results.db $O/${TAG}_kernel_stats_serial.txt "serial" 7
