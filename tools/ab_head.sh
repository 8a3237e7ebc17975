#!/bin/bash
# Development aid: build the library of a git revision (default HEAD) as tools/bin/libendo_hip_<name>.so, for same-box A/B runs against the
# working tree through ENDO_HIP_LIB (tools/ab_config.sh).   usage: tools/ab_head.sh [name] [rev]
set -e
name=${1:-head}; rev=${2:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
work=/tmp/variant_$name
rm -rf $work; mkdir -p $work/pkg/csrc $work/include
for f in $(git -C $root ls-tree --name-only $rev endoscopydepthestimation-pytorch_amd/csrc/); do git -C $root show $rev:$f > $work/pkg/csrc/$(basename $f); done
git -C $root show $rev:include/endo_hip.h > $work/include/endo_hip.h
cd $work/pkg/csrc
for f in *.hip; do
    extra=""; [ $f = dgrad_wino3.hip ] && extra="-fno-slp-vectorize"
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC $extra -c $f -o $work/${f%.hip}.o &
done
wait
mkdir -p $root/tools/bin
hipcc --offload-arch=gfx950 -shared -o $root/tools/bin/libendo_hip_$name.so $work/*.o
echo built $root/tools/bin/libendo_hip_$name.so
