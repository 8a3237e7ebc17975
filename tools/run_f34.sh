cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_d2
python bench.py > gpurun_out/r04_d2/r04_d_bench.json 2> gpurun_out/r04_d2/bench.err
python bench.py --config 3 --no-cpu-baseline > gpurun_out/r04_d2/r04_d_bench_config3.json 2> gpurun_out/r04_d2/bench3.err
python bench.py --kernel-option 0=5 --no-cpu-baseline > gpurun_out/r04_d2/r04_d_bench_option_wino4_fwd.json 2> gpurun_out/r04_d2/bench5.err
tail -1 gpurun_out/r04_d2/r04_d_bench.json | cut -c1-400
