for s in 1,3,2160,3840 32,3,1080,1920; do
  for f in -1 2 3; do echo "== shape $s PB_FFT_LOGNB=$f"; PB_FFT_LOGNB=$f python tools/bench_estimate.py --shape $s --reps 30 2>&1 | tail -5 | grep "cols\|rows"; done
done
