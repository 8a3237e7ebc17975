cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 ) > gpurun_out/r4g/pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r4g/pytest.txt
timeout 600 python bench.py > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err
tail -24 gpurun_out/r4g/pytest.txt
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4g/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline_depth_warp"], d["depth_warp_fwd_bwd_ms_per_pair"])
PY
