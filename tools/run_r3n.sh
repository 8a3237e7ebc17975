mkdir -p gpurun_out/r3n
R=$GRAFT_REPO_ROOT
python tools/bf16_conv_bench.py > gpurun_out/r3n/bench.txt 2>&1
cat gpurun_out/r3n/bench.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/sq$i -o s -- python3 $R/tools/bf16_conv_bench.py 3 > /dev/null 2> $R/gpurun_out/r3n/sq$i.err
done
python3 $R/tools/pmc_sq_report.py $R/gpurun_out/r3n/sq_counters.txt /tmp/sq1/s_results.db /tmp/sq2/s_results.db /tmp/sq3/s_results.db
grep -A2 "bf16_conv_kernel" $R/gpurun_out/r3n/sq_counters.txt | cut -c1-330
