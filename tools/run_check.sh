cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
timeout 900 python -m pytest tests/test_reader.py -m gpu -q -x 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "warp_consistency or loss_head or train_step_golden or gap_scaled" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4n/bench.json 2> gpurun_out/r4n/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4n/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline_depth_warp"]["device_ms_per_pair"], d["roofline_depth_warp"]["frac"])
PY
