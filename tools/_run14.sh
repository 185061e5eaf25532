python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_estimation_paths.py tests/test_gpu_round5_forms.py -x -q -k "halo or cfg3 or half_gradient or default_threshold or variants or extra" 2>&1 | tail -6
python tools/run_configs.py 2>&1 | grep "cfg3:" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        p = json.loads(l); print(p['config'][:40], p['ms'], p['mp_per_s'])
"
