#!/bin/bash
# usage: ab_only.sh <outdir> <rounds> <names...>
O=gpurun_out/$1; mkdir -p $O; rounds=$2; shift 2
bash tools/ab.sh $rounds $O/ab.txt "" "$@"; cat $O/ab.txt
