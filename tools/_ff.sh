python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for c in cfg2 cfg3 cfg4 cfg5; do for v in 0 -1 0 -1; do PB_FFT_FIRST=$v python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-context 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);s=d['stages_ms_per_step'];print('$c first=$v',d['ms_per_step'],{k:s[k] for k in s if 'grad' in k},d['parity']['max_abs'] if d.get('parity') else None)"; done; done
python tools/bench_small_class.py
PB_FFT_FIRST=0 python tools/bench_small_class.py
