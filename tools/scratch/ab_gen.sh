#!/bin/bash
# usage: ab_gen.sh <outdir> <rounds> <names...>; then the full GPU suite
O=gpurun_out/$1; mkdir -p $O; rounds=$2; shift 2
bash tools/ab.sh $rounds $O/ab.txt "" "$@"; cat $O/ab.txt
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $O/t.txt; cat $O/t.txt
