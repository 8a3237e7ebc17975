cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f34f
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kernel_forms or full_size" 2>&1 | tail -2
for i in 1 2; do
 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/f34f/bench_new_$i.json 2>> gpurun_out/f34f/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/f34f/bench_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, round(d["value"],1), round(d["ms_per_step"],3), {k: round(v,2) for k,v in d["roofline_serial"]["families_ms_per_step"].items()})
PY
