for s in 1,3,2160,3840 1,3,1080,1920 32,3,1080,1920; do
  for f in 1 2 3; do echo "== shape $s COLS_FIXED=$f"; PB_COLS_FIXED=$f python tools/bench_estimate.py --shape $s --reps 50 2>&1 | tail -5 | grep "cols"; done
done
PB_COLS_FIXED=2 python -m pytest tests/test_gpu_estimation_paths.py -x -q -k fixed_plan_columns 2>&1 | tail -2
