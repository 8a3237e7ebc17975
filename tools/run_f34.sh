cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f34g
for i in 1 2; do
 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-option 7=2 > gpurun_out/f34g/bench_half_$i.json 2>> gpurun_out/f34g/bench.err
 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/f34g/bench_full_$i.json 2>> gpurun_out/f34g/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/f34g/bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, round(d["value"],1), round(d["ms_per_step"],3), {k: round(v,2) for k,v in d["roofline_serial"]["families_ms_per_step"].items()})
    except Exception as e: print(f, "ERR", e)
PY
