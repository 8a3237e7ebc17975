#!/bin/bash
# Build a variant libendo_hip from a modified COPY of csrc (development aid for in-job A/B runs with ENDO_HIP_LIB).
#   tools/build_variant.sh <name> <sed-script-or-empty> [file=git-rev ...]
# e.g. tools/build_variant.sh b1 "" wgrad_taps_kernels.h=HEAD~3
set -e
name=$1; sedscript=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
work=/tmp/variant_$name
rm -rf $work; mkdir -p $work/pkg/csrc $work/include
cp $root/endoscopydepthestimation-pytorch_amd/csrc/* $work/pkg/csrc/
cp $root/include/endo_hip.h $work/include/
for spec in "$@"; do
    f=${spec%%=*}; rev=${spec#*=}
    git -C $root show $rev:endoscopydepthestimation-pytorch_amd/csrc/$f > $work/pkg/csrc/$f
done
if [ -n "$sedscript" ]; then sed -i -E "$sedscript" $work/pkg/csrc/*.h $work/pkg/csrc/*.hip; fi
cd $work/pkg/csrc
for f in geometry losses optimizer prof scatter net; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -c $f.hip -o $work/$f.o &
done
wait
mkdir -p $root/tools/bin
hipcc --offload-arch=gfx950 -shared -o $root/tools/bin/libendo_hip_$name.so $work/geometry.o $work/losses.o $work/optimizer.o $work/prof.o $work/scatter.o $work/net.o
echo built $root/tools/bin/libendo_hip_$name.so
