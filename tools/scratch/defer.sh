#!/bin/bash
O=gpurun_out/r6x; mkdir -p $O
run() { # label, env..., extra args
  label=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA > /tmp/o.json 2>/tmp/e.txt
  python - "$label" >> $O/defer.txt <<'PY'
import json,sys
d=json.loads([l for l in open('/tmp/o.json') if l.startswith('{')][-1])
print("%-24s %8.2f frame-pairs/s %7.3f ms/step" % (sys.argv[1], d["value"], d["ms_per_step"]))
PY
}
for r in 1 2; do
  EXTRA=""
  run "defer 0" ENDO_WGRAD_DEFER=0
  run "defer 1" ENDO_WGRAD_DEFER=1
  run "defer 1 + TU" ENDO_WGRAD_DEFER=1 ENDO_WGRAD_DEFER_TU=1
  run "defer 2 + TU" ENDO_WGRAD_DEFER=2 ENDO_WGRAD_DEFER_TU=1
  EXTRA="--config 3"
  run "config 3: defer 0" ENDO_WGRAD_DEFER=0
  run "config 3: defer 1" ENDO_WGRAD_DEFER=1
  run "config 3: defer 2 + TU" ENDO_WGRAD_DEFER=2 ENDO_WGRAD_DEFER_TU=1
done
cat $O/defer.txt
