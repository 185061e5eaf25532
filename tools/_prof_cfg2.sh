bash tools/profile_bench.sh r06 cfg2 > gpurun_out/r06_profile_cfg2.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -2 gpurun_out/r06_profile_cfg2.log
