#!/bin/bash
# In-job A/B of library variants (development aid): tools/ab_bench.sh <rounds> <name> [<name> ...]; "main" = the in-tree library
rounds=$1; shift
for r in $(seq $rounds); do
  for name in "$@"; do
    if [ "$name" = main ]; then lib=""; else lib=$PWD/tools/bin/libendo_hip_$name.so; fi
    ENDO_HIP_LIB=$lib python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline > /tmp/ab.out 2> /tmp/ab.err
    python - "$name" <<'PY'
import json, sys
line = [l for l in open('/tmp/ab.out') if l.startswith('{')][-1]
d = json.loads(line)
b = json.loads([l for l in open('/tmp/ab.err') if l.startswith('{')][-1])["family_breakdown_one_step"]
print("%-6s %7.3f ms/step  fwd %.3f dgrad %.3f wgrad %.3f | up_fwd %.3f pool_fwd %.3f dgrad_o %.3f wgrad_o %.3f small %.3f" % (
    sys.argv[1], d["ms_per_step"], b["conv3x3_dense_fwd"]["ms"], b["dgrad_dense"]["ms"], b["wgrad_dense"]["ms"],
    b["conv3x3_up_fwd"]["ms"], b["conv1x1_pool_fwd"]["ms"], b["dgrad_other"]["ms"], b["wgrad_other"]["ms"], b["small"]["ms"]))
PY
  done
done
