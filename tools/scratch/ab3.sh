#!/bin/bash
# usage: ab3.sh <outdir> <names...>: A/B at configs 1 and 3, then the parity-relevant part of the GPU suite
O=gpurun_out/$1; mkdir -p $O; shift
bash tools/ab.sh 2 $O/ab.txt "" "$@"; cat $O/ab.txt
bash tools/ab.sh 1 $O/ab3.txt "--config 3" "$@"; cat $O/ab3.txt
timeout 1500 python -m pytest tests -q -m gpu -x -k "not storage and not bf16 and not fp16" 2>&1 | tail -3
