#!/usr/bin/env python3
"""Golden vectors from the REFERENCE itself for `edgetaping=True` with a ker_size ABOVE 25 (this container only; same import
recipe as make_golden.py).  edgetaper_alpha (edgetaper.py:10-23) takes a kernel of any size: its weights are circular
autocorrelations of the kernel's projections.  Sizes above 25 take the large-kernel pass of the engine
(csrc/conv_big.hip).  Two links per (size, method):
x1 = f(x0), x2 = f(x1) with f = one reference iteration (even sizes amplify rounding differences from iteration to iteration).

    python tests/golden/make_golden_big_taper.py      # writes tests/golden/pipeline_big_taper.npz
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
from polyblur import polyblur_deblurring  # noqa: E402
from polyblur_amd.synthetic import synthetic_blurry_batch  # noqa: E402

KW = dict(c=0.362, b=0.468, alpha=6, beta=1, edgetaping=True)
x, _ = synthetic_blurry_batch(1, 3, 110, 150, seed0=6363)
d = {"x0": x}
for k in (31, 48):
    for method in ("fft", "direct"):
        cur = x
        for it in range(2):
            cur = polyblur_deblurring(torch.from_numpy(cur.copy()), n_iter=1, ker_size=k, method=method, **KW).numpy()
            d["k%d_%s_x%d" % (k, method, it + 1)] = cur
np.savez_compressed(os.path.join(HERE, "pipeline_big_taper.npz"), **d)
print({k: v.shape for k, v in d.items()})
