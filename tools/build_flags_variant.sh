#!/bin/bash
# Build a variant libendo_hip with extra compiler flags (development aid for in-job A/B runs with ENDO_HIP_LIB).
#   tools/build_flags_variant.sh <name> "<extra hipcc flags>"
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
work=/tmp/variant_$name
rm -rf $work; mkdir -p $work
cd $root/endoscopydepthestimation-pytorch_amd/csrc
objs=""
for f in geometry losses optimizer prof scatter head jpeg filter dgrad_wino3 net16 net16h net; do
    fx=""; [ $f = dgrad_wino3 ] && fx="-fno-slp-vectorize"
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC $extra $fx -c $f.hip -o $work/$f.o &
    objs="$objs $work/$f.o"
done
wait
mkdir -p $root/tools/bin
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/bin/libendo_hip_$name.so $objs
echo built $root/tools/bin/libendo_hip_$name.so
