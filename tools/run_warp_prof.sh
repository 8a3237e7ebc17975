cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4i
cd $R && timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "warp_consistency or test_losses or loss_head or train_step_golden or nonfinite" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x -k "gradient_scale or against_oracle" 2>&1 | tail -3
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/wc -o wc -- python3 $R/tools/warp_consistency_workload.py > /dev/null 2> $R/gpurun_out/r4i/prof.err
python3 $R/tools/summarize_rocprof.py /tmp/wc/wc_results.db $R/gpurun_out/r4i/warp_consistency_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python3 tools/warp_consistency_workload.py" 50
head -12 $R/gpurun_out/r4i/warp_consistency_kernel_stats.txt | cut -c1-170
cd $R; python bench.py --no-cpu-baseline > gpurun_out/r4i/bench.json 2> gpurun_out/r4i/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4i/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline_depth_warp"])
PY
