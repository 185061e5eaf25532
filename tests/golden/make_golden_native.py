"""Golden vectors from the reference's OWN native domain-transform code (NC.cpp, RF.cpp), compiled
as CPU torch extensions by oracle/build_ref_native.py from /root/reference (this container only).
Writes tests/golden/native_dt.npz: inputs + the reference's outputs.  Run:  python tests/golden/make_golden_native.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.build_ref_native import build                       # noqa: E402
from polyblur_amd.synthetic import synthetic_blurry_batch      # noqa: E402

nc = build("NC")
rf = build("RF")
rng = np.random.default_rng(20260929)
d = {}
cases = {
    "a": (rng.random((1, 3, 12, 16), dtype=np.float32), 2.0, 0.8, 1),
    "b": (rng.random((1, 3, 23, 31), dtype=np.float32), 60.0, 0.4, 3),
    "c": (synthetic_blurry_batch(1, 3, 48, 64, seed0=11)[0], 2.0, 0.8, 1),
    "d": (synthetic_blurry_batch(1, 3, 40, 36, seed0=12)[0], 8.0, 0.3, 3),
    "e": (synthetic_blurry_batch(1, 3, 33, 70, seed0=13)[0], 3.0, 0.1, 2),
}
for k, (x, ss, sr, n) in cases.items():
    d["x_" + k] = x
    d["p_" + k] = np.asarray([ss, sr, n], np.float64)
    d["nc_" + k] = nc.normalized_convolution(torch.from_numpy(x.copy()), ss, sr, n).numpy()
    d["rf_" + k] = rf.recursive_filter(torch.from_numpy(x.copy()), ss, sr, n).numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "native_dt.npz"), **d)
print({k: v.shape for k, v in d.items()})
