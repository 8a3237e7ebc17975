#!/bin/bash
# In-job A/B of kernel options on one box (development aid): tools/ab_options.sh <rounds> "<label>:<opts>" ...; opts = "" or "1=2,3=512"
rounds=$1; shift
for r in $(seq $rounds); do
  for spec in "$@"; do
    label=${spec%%:*}; opts=${spec#*:}
    flags=""
    IFS=',' read -ra arr <<< "$opts"
    for o in "${arr[@]}"; do [ -n "$o" ] && flags="$flags --kernel-option $o"; done
    python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline $flags > /tmp/ab.out 2> /tmp/ab.err
    python - "$label" <<'PY'
import json, sys
line = [l for l in open('/tmp/ab.out') if l.startswith('{')][-1]
d = json.loads(line)
b = json.loads([l for l in open('/tmp/ab.err') if l.startswith('{')][-1])["family_breakdown_one_step"]
print("%-10s %7.3f ms/step  fwd %.3f dgrad %.3f wgrad %.3f | up_fwd %.3f pool_fwd %.3f dgrad_o %.3f wgrad_o %.3f small %.3f" % (
    sys.argv[1], d["ms_per_step"], b["conv3x3_dense_fwd"]["ms"], b["dgrad_dense"]["ms"], b["wgrad_dense"]["ms"],
    b["conv3x3_up_fwd"]["ms"], b["conv1x1_pool_fwd"]["ms"], b["dgrad_other"]["ms"], b["wgrad_other"]["ms"], b["small"]["ms"]))
PY
  done
done
