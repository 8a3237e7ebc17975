# The profile set committed under profiles/ (run on an MI355X box through gpurun): tools/final_profiles.sh <tag>, e.g. r03_f
set -x
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
python3 $R/tools/summarize_rocprof.py /tmp/kt/kt_results.db $O/${TAG}_kernel_stats.txt "$CMD" 7
python3 $R/tools/summarize_rocprof.py --by-grid /tmp/kt/kt_results.db $O/${TAG}_kernel_stats_by_grid.txt
python3 $R/tools/gpu_busy.py /tmp/kt/kt_results.db > $O/${TAG}_gpu_busy.txt 2>&1
python3 $R/tools/chain_trace.py /tmp/kt/kt_results.db 2 > $O/${TAG}_chain_trace.txt 2>&1
python3 $R/tools/stream_timeline.py /tmp/kt/kt_results.db > $O/${TAG}_stream_timeline.txt 2>&1
# the same trace with the weight gradients in line (every kernel alone on the chip): stand-alone kernel times
rocprofv3 --kernel-trace --stats -d /tmp/kts -o kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel-option 5=0 > /dev/null 2> $O/kts.err
python3 $R/tools/summarize_rocprof.py /tmp/kts/kt_results.db $O/${TAG}_kernel_stats_serial.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel-option 5=0   (ENDO_OPT_WGRAD_OVERLAP = 0: weight gradients in line, every kernel alone on the chip)" 7
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o f -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o w -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pw.err
python3 $R/tools/pmc_traffic.py /tmp/pf/f_results.db /tmp/pw/w_results.db $O/${TAG}_pmc_traffic.json 7 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline" > $O/pmc_traffic.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/sq$i -o s -- python3 $R/tools/pmc_workload.py 1 > /dev/null 2> $O/sq$i.err
done
python3 $R/tools/pmc_sq_report.py $O/${TAG}_sq_counters.txt /tmp/sq1/s_results.db /tmp/sq2/s_results.db /tmp/sq3/s_results.db > $O/sq.log 2>&1
cd $R
# the bench line reads profiles/*_pmc_traffic.json and *_kernel_stats.json of THIS source revision: put them in place first
cp $O/${TAG}_pmc_traffic.json $O/${TAG}_kernel_stats.json $R/profiles/
python bench.py > $O/${TAG}_bench.json 2> $O/bench.err
# the bf16-storage mode (bench.py --config 2): kernel trace, HBM traffic of its families, its bench line
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt -- python3 $R/bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_config2_under_rocprof.json 2> $O/kt2.err
python3 $R/tools/summarize_rocprof.py /tmp/kt2/kt_results.db $O/${TAG}_kernel_stats_config2.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline" 7
python3 $R/tools/summarize_rocprof.py --by-grid /tmp/kt2/kt_results.db $O/${TAG}_kernel_stats_config2_by_grid.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf2 -o f -- python3 $R/bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pf2.err
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw2 -o w -- python3 $R/bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/pw2.err
python3 $R/tools/pmc_traffic.py /tmp/pf2/f_results.db /tmp/pw2/w_results.db $O/${TAG}_pmc_traffic_config2.json 7 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline" > $O/pmc_traffic2.log 2>&1
cd $R
cp $O/${TAG}_pmc_traffic_config2.json $O/${TAG}_kernel_stats_config2.json $R/profiles/
python bench.py --config 2 --no-cpu-baseline > $O/${TAG}_bench_config2.json 2> $O/bench2.err
python bench.py --config 4 --no-cpu-baseline > $O/${TAG}_bench_config4.json 2> $O/bench4.err
python bench.py --config 3 --no-cpu-baseline > $O/${TAG}_bench_config3.json 2> $O/bench3.err
tail -1 $O/${TAG}_bench.json | cut -c1-900
ls -la $O
# the F(3x3, 4x4) weight gradient stand-alone against the direct kernels (tools/x3_bench, built by hand: see its header) and its counters
if [ -x $R/tools/bin/x3_bench ]; then
  ( for a in "48 16 256 320" "60 16 256 320" "84 16 256 320" "128 16 256 320" "180 16 256 320" "144 16 128 160" "228 16 128 160" "156 16 64 80" "264 16 64 80"; do
      $R/tools/bin/x3_bench $a 2>&1 | grep "dense-layer\|fp32 MFMA (v\|F(3x3" | cut -c1-240; done ) > $O/${TAG}_wgrad_f34_bench.txt
  cd /tmp
  i=0
  for set in "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    rocprofv3 --pmc $set -d /tmp/f34p$i -o s -- $R/tools/bin/x3_bench 180 16 256 320 > /dev/null 2> $O/f34p$i.err
  done
  python3 $R/tools/pmc_sq_report.py $O/f34_all.txt /tmp/f34p*/s_results.db > $O/f34sq.log 2>&1
  ( echo "# rocprofv3 --pmc <set> -- tools/bin/x3_bench 180 16 256 320 (five passes; values per dispatch): the F(3x3, 4x4) kernel, its diagnostic build without activation loads (<2>), the direct n-split kernel (<3, 0, 0>)"; grep -A3 "f34_kernel\|nsplit_kernel<3, 0, 0>" $O/f34_all.txt | cut -c1-1600 ) > $O/${TAG}_wgrad_f34_sq_counters.txt
  cd $R
fi
