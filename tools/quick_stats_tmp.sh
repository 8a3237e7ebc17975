#!/bin/bash
# quick kernel-time picture of the current tree: overlapped and one-at-a-time (development aid)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_kt.json 2> $O/kt.err
python3 $R/tools/summarize_rocprof.py /tmp/kt/kt_results.db $O/kernel_stats.txt "overlapped" 7 > $O/sum.log 2>&1
python3 $R/tools/stream_timeline.py /tmp/kt/kt_results.db > $O/stream_timeline.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/kts -o kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel-option 5=0 > $O/bench_kts.json 2> $O/kts.err
python3 $R/tools/summarize_rocprof.py /tmp/kts/kt_results.db $O/kernel_stats_serial.txt "serial" 7 > $O/sum2.log 2>&1
python3 $R/tools/summarize_rocprof.py --by-grid /tmp/kts/kt_results.db $O/kernel_stats_serial_by_grid.txt > $O/sum3.log 2>&1
