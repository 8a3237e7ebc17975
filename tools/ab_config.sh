#!/bin/bash
# Same-box A/B of the working tree's library against tools/bin/libendo_hip_<name>.so (tools/ab_head.sh): alternating bench.py runs.
#   usage (on the GPU box): tools/ab_config.sh <rounds> <config> <name>
rounds=$1; config=$2; name=$3
for r in $(seq $rounds); do
  for v in tree $name; do
    if [ $v = tree ]; then unset ENDO_HIP_LIB; else export ENDO_HIP_LIB=$PWD/tools/bin/libendo_hip_$name.so; fi
    python bench.py --config $config --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-6s %.3f ms/step  %.1f frame-pairs/s' % ('$v', d['ms_per_step'], d['value']))"
  done
done
