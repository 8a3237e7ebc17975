#!/bin/bash
# Everything a round commits under profiles/ in ONE gpurun call: tools/final_all.sh <tag>
#   the profile set (tools/final_profiles.sh), the round's stand-alone kernel benches (tools/bin/*, built by hand: see each tool's header),
#   the GPU test suite and __graft_entry__.smoke()
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
bash $R/tools/final_profiles.sh $TAG > $O/final_profiles.log 2>&1
cd $R
{
  echo "# tools/bin/wino_bench <C0> 16 <h> <w> 1: the dense block's base-channel data gradient stand-alone (direct / F(2x2,3x3) per tile / persistent blocks)"
  for a in "144 16 256 320 1" "48 16 256 320 1" "96 16 128 160 1" "192 16 128 160 1"; do timeout 300 tools/bin/wino_bench $a 2>&1 | grep -v "^  worker" | cut -c1-220; done
} > $O/${TAG}_wino3p_bench.txt 2>&1
{
  echo "# tools/bin/td_bench <C> 16 <h> <w> [1 = forward]: the transition-down 1x1 convolution's data gradient and forward, per tile against persistent blocks"
  for a in "96 16 256 320" "144 16 128 160" "96 16 256 320 1" "144 16 128 160 1"; do timeout 300 tools/bin/td_bench $a 2>&1 | cut -c1-220; done
} > $O/${TAG}_td_bench.txt 2>&1
{
  echo "# tools/bin/tdw_bench <C> 16 <h> <w>: the transition-down weight gradient, round 5's dword-DMA kernel against the 16-byte-DMA kernel in its two block shapes"
  for a in "96 16 256 320" "144 16 128 160" "192 16 64 80" "240 16 32 40"; do timeout 120 tools/bin/tdw_bench $a 2>&1 | cut -c1-220; done
} > $O/${TAG}_tdw_bench.txt 2>&1
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 ) > $O/${TAG}_pytest_gpu_tail.txt 2>&1
cat $O/${TAG}_pytest_gpu_tail.txt
tail -1 $O/${TAG}_bench.json | cut -c1-400
