#!/bin/bash
O=gpurun_out/r6ad; mkdir -p $O
run() { label=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/o.json 2>/tmp/e.txt
  python - "$label" >> $O/w2.txt <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
rs=d["roofline_serial"]["families_ms_per_step"]
print("%-28s %8.2f frame-pairs/s %7.3f ms/step | stand-alone fwd %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], rs["conv3x3_dense_fwd"]))
PY
}
for r in 1 2; do
  run "wino2 from level 1 (default)" ENDO_X=1
  run "wino2 also at level 2" ENDO_WINO2_MIN=512
  run "wino2 also at levels 2-3" ENDO_WINO2_MIN=128
done
cat $O/w2.txt
ENDO_WINO2_MIN=128 timeout 600 python -m pytest tests -q -m gpu -x -k "fixture or training_step or winograd" 2>&1 | tail -3
