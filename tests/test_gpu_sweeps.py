"""Short runs of the random sweeps of tools/ (lab equipment: each compares the engine with the oracle on seeded random cases and
prints one summary line).  The long runs are recorded in profiles/r06_random_sweeps.txt; these keep a slice of each in the suite --
the sweeps are how round 6 found the job-list bug behind an edgetaper, the orientation of caller-supplied kernels under the wrap
boundary and the patch lattice without a patch."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), *[str(a) for a in args]], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    return lines[-1], lines


@pytest.mark.parametrize("script,args,pattern", [
    ("sweep_random.py", (24, 64), r": 0 outside tolerance"),
    ("sweep_random_taper.py", (0, 30), r": 0 outside tolerance"),
    ("sweep_random_kernels.py", (0, 30), r": 0 outside tolerance"),
    ("sweep_random_patches.py", (0, 16), r": 0 outside 1e-4"),
    ("sweep_random_batch.py", (0, 16), r": 0 with an image that differs"),
    ("sweep_random_dt.py", (0, 40), r": 0 outside tolerance or differing"),
    ("sweep_random_stages.py", (0, 30), r": 0 outside tolerance"),
    ("sweep_random_sequence.py", (120, 7), r": 0 outside tolerance"),
])
def test_sweep_slice(script, args, pattern):
    last, lines = _run(script, *args)
    summary = [l for l in lines if re.search(r"cases \d+\.\.\d+|sequence of \d+ calls", l)]
    assert summary and re.search(pattern, summary[0]), "\n".join(lines[-12:])
