#!/bin/bash
# usage: ab_c2.sh <outdir> <rounds> <names...>: A/B of the bf16-storage step, then the 16-bit tests
O=gpurun_out/$1; mkdir -p $O; rounds=$2; shift 2
bash tools/ab.sh $rounds $O/ab.txt "--config 2" "$@"; cat $O/ab.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "bf16 or fp16 or storage or s16 or half or 16" 2>&1 | tail -4 > $O/t.txt; cat $O/t.txt
