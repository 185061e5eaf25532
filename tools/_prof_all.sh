python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r06_final_tests.txt
for c in cfg2 cfg3 cfg4 cfg5; do bash tools/profile_bench.sh r06 $c > gpurun_out/r06_profile_$c.log 2>&1; done
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
for c in cfg3 cfg4 cfg5; do python bench.py --config $c --steps 5 --warmup 2 --no-context > gpurun_out/r06_bench_$c.json 2> gpurun_out/r06_bench_$c.err; done
cat gpurun_out/r06_final_tests.txt
