#!/usr/bin/env python3
"""More golden vectors from the REFERENCE itself (this container only; same import recipe as make_golden.py):
non-default interpolation grids (n_interpolated_angles -- the `.long()`-truncated grids and the Keys
interpolation matrix of other sizes), the module's own defaults at n_iter=3, and two more image sizes.

    python tests/golden/make_golden_extra.py      # writes tests/golden/pipeline_extra.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sk = types.ModuleType("skimage")
sk.img_as_float32 = lambda x: np.asarray(x, np.float32) / (255.0 if np.asarray(x).dtype == np.uint8 else 1.0)
sys.modules["skimage"] = sk
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

import torch  # noqa: E402

torch.set_num_threads(8)
from polyblur import polyblur_deblurring, PolyblurDeblurring  # noqa: E402
from polyblur_amd.synthetic import synthetic_blurry_batch  # noqa: E402

KW = dict(c=0.362, b=0.468, alpha=6, beta=1)
x, _ = synthetic_blurry_batch(1, 3, 96, 130, seed0=4242)
y, _ = synthetic_blurry_batch(2, 3, 61, 77, seed0=4343)              # odd sizes, batch
d = {"x": x, "y": y}
# (n_angles != 6 raises inside the reference -- shape mismatch in cubic_interpolator -- so only the interpolation grid varies)
cases = {
    "a6_i45": dict(n_iter=2, n_angles=6, n_interpolated_angles=45),
    "a6_i12": dict(n_iter=2, n_angles=6, n_interpolated_angles=12),
    "a6_i60": dict(n_iter=1, n_angles=6, n_interpolated_angles=60),
    "a6_i7": dict(n_iter=2, n_angles=6, n_interpolated_angles=7),
}
for name, kw in cases.items():
    for method in ("fft", "direct"):
        d["%s_%s" % (name, method)] = polyblur_deblurring(torch.from_numpy(x.copy()), method=method, **KW, **kw).numpy()
d["odd_batch_fft"] = polyblur_deblurring(torch.from_numpy(y.copy()), n_iter=3, method="fft", **KW).numpy()
d["module_defaults_n3"] = PolyblurDeblurring()(torch.from_numpy(x.copy()), n_iter=3).numpy()
d["functional_defaults_n2"] = polyblur_deblurring(torch.from_numpy(x.copy()), n_iter=2).numpy()
np.savez_compressed(os.path.join(HERE, "pipeline_extra.npz"), **d)
print({k: v.shape for k, v in d.items()})
