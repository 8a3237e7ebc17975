#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6aa; mkdir -p $O
for a in "96 16 256 320" "144 16 128 160" "96 16 256 320 1" "144 16 128 160 1"; do timeout 300 tools/bin/td_bench $a 2>&1 | grep "transition\|persistent\|per tile" | cut -c1-230; done > $O/td.txt 2>&1
cat $O/td.txt
bash tools/ab.sh 2 $O/ab.txt "" il2 main; cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o f -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o w -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pw.err
python3 $R/tools/pmc_traffic.py /tmp/pf/f_results.db /tmp/pw/w_results.db $O/il_pmc_traffic.json 7 "interleaved tile walk" > $O/pmc_traffic.log 2>&1
tail -3 $O/pmc_traffic.log
cd $R; timeout 900 python -m pytest tests -q -m gpu -x -k "persistent or fixture or transparent or training_step or transition" 2>&1 | tail -3
