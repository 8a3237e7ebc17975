set -x
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o f -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o w -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pw.err
python3 $R/tools/pmc_traffic.py /tmp/pf/f_results.db /tmp/pw/w_results.db $O/${TAG}_pmc_traffic.json 7 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline" > $O/pmc_traffic.log 2>&1
cd $R
cp $O/${TAG}_pmc_traffic.json $O/${TAG}_kernel_stats.json $R/profiles/ 2>/dev/null
cp $R/gpurun_out/r06_a/r06_a_kernel_stats.json $R/profiles/ 2>/dev/null
python bench.py > $O/${TAG}_bench.json 2> $O/bench.err
python -m pytest tests -q -m gpu 2>&1 | tail -2 > $O/${TAG}_pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" >> $O/${TAG}_pytest_gpu_tail.txt 2>&1
tail -3 $O/pmc_traffic.log
