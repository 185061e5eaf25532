echo "# tools/sweep_fixed_plans.py 0 40; sweep_random.py 24 200; sweep_random_large.py 0 30; sweep_random_io.py; sweep_random_sizes.py 0 120 at the round-6 sources: every case against the oracle"
python tools/sweep_fixed_plans.py 0 40 2>&1 | grep -v amdgpu.ids
python tools/sweep_random.py 24 200 2>&1 | grep -v amdgpu.ids
python tools/sweep_random_large.py 0 30 2>&1 | grep -v amdgpu.ids
python tools/sweep_random_io.py 2>&1 | grep -v amdgpu.ids
python tools/sweep_random_sizes.py 0 120 2>&1 | grep -v amdgpu.ids
