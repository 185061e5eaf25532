for s in 1,3,2160,3840 1,3,1080,1920 32,3,1080,1920; do
  for f in "1 1" "2 1" "1 2" "2 2"; do set -- $f; echo "== shape $s COLS_FIXED=$1 ROWS_FIXED=$2"; PB_COLS_FIXED=$1 PB_ROWS_FIXED=$2 python tools/bench_estimate.py --shape $s --reps 40 2>&1 | tail -5 | grep "cols\|rows"; done
done
for f in "1 1" "2 2"; do set -- $f; echo "== 8K f16 COLS_FIXED=$1 ROWS_FIXED=$2"; PB_COLS_FIXED=$1 PB_ROWS_FIXED=$2 python tools/bench_estimate.py --shape 1,3,4320,7680 --dtype f16 --reps 20 2>&1 | tail -5 | grep "cols\|rows"; done
PB_COLS_FIXED=2 PB_ROWS_FIXED=2 python -m pytest tests/test_gpu_estimation_paths.py -x -q 2>&1 | tail -3
