# config-2 (bf16 storage) step timeline: spans per phase, per-stream busy time in the backward, GPU idle
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_ao; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt -- python3 $R/bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/kt.err
python3 $R/tools/stream_timeline.py /tmp/kt2/kt_results.db > $O/r03_ao_stream_timeline_config2.txt 2>&1
python3 $R/tools/gpu_busy.py /tmp/kt2/kt_results.db > $O/r03_ao_gpu_busy_config2.txt 2>&1
tail -40 $O/r03_ao_stream_timeline_config2.txt
tail -25 $O/r03_ao_gpu_busy_config2.txt
