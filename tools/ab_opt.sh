#!/bin/bash
# in-job A/B of one kernel option: tools/ab_opt.sh <rounds> <out> "<args A>" "<args B>"
rounds=$1; out=$2; A=$3; B=$4
for r in $(seq $rounds); do
  for v in A B; do
    if [ $v = A ]; then extra=$A; else extra=$B; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline $extra > /tmp/ab.out 2> /tmp/ab.err
    python - "$v [$extra]" >> $out <<'PY'
import json, sys
line = [l for l in open('/tmp/ab.out') if l.startswith('{')][-1]
d = json.loads(line)
rs = (d.get("roofline_serial") or {}).get("families_ms_per_step") or {}
print("%-28s %8.2f frame-pairs/s %7.3f ms/step | stand-alone fwd %.3f dgrad %.3f wgrad %.3f | in-step %s %.3f ms" % (
    sys.argv[1], d["value"], d["ms_per_step"], rs.get("conv3x3_dense_fwd", 0), rs.get("dgrad_dense", 0), rs.get("wgrad_dense", 0),
    d["roofline"]["kernel"], d["roofline"].get("family_ms_per_step") or 0))
PY
  done
done
