#!/bin/bash
# In-job A/B of library builds (development aid): tools/ab.sh <rounds> <outfile> <bench args or ""> <name> [<name> ...]
#   "main" = the in-tree library, any other name = tools/bin/libendo_hip_<name>.so (ENDO_HIP_LIB); alternates the builds on ONE box
rounds=$1; out=$2; extra=$3; shift 3
for r in $(seq $rounds); do
  for name in "$@"; do
    if [ "$name" = main ]; then lib=""; else lib=$PWD/tools/bin/libendo_hip_$name.so; fi
    ENDO_HIP_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline $extra > /tmp/ab.out 2> /tmp/ab.err
    python - "$name" >> $out <<'PY'
import json, sys
line = [l for l in open('/tmp/ab.out') if l.startswith('{')][-1]
d = json.loads(line)
rs = (d.get("roofline_serial") or {}).get("families_ms_per_step") or {}
print("%-6s %8.2f frame-pairs/s %7.3f ms/step | stand-alone fwd %.3f dgrad %.3f wgrad %.3f | in-step %s %.3f ms" % (
    sys.argv[1], d["value"], d["ms_per_step"], rs.get("conv3x3_dense_fwd", 0), rs.get("dgrad_dense", 0), rs.get("wgrad_dense", 0),
    d["roofline"]["kernel"], d["roofline"].get("family_ms_per_step") or 0))
PY
  done
done
