#!/bin/bash
# chain trace + timeline of the bf16-storage step (config 2), overlapped and serial
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt -- python3 $R/bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/b.json 2> $O/kt2.err
python3 $R/tools/chain_trace.py /tmp/kt2/kt_results.db 2 > $O/c2_chain_trace.txt 2>&1
python3 $R/tools/stream_timeline.py /tmp/kt2/kt_results.db > $O/c2_stream_timeline.txt 2>&1
python3 $R/tools/gpu_busy.py /tmp/kt2/kt_results.db > $O/c2_gpu_busy.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/kts2 -o kt -- python3 $R/bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --kernel-option 5=0 > $O/bs.json 2> $O/kts2.err
python3 $R/tools/summarize_rocprof.py /tmp/kts2/kt_results.db $O/c2_kernel_stats_serial.txt "config 2, --kernel-option 5=0" 7
python3 $R/tools/summarize_rocprof.py --by-grid /tmp/kts2/kt_results.db $O/c2_kernel_stats_serial_by_grid.txt
tail -5 $O/c2_stream_timeline.txt
