#!/usr/bin/env python3
"""Golden vectors from the REFERENCE itself for the ill-conditioned corner: an EVEN ker_size over three iterations (this
container only; same import recipe as make_golden.py).

An even size is the Gaussian on the off-centre grid arange(k) - (k - 1) // 2 (blur_estimation.py:221-223); under 'fft' the
kernel array is rolled by k // 2 (filters.py:255-273), so its transform carries a half-sample phase and the polynomial
reaches |a3| + |a2| + |a1| + |b| near Nyquist: every iteration multiplies the rounding differences of the earlier ones
(1e-5 after two iterations -> 8e-5 after three).  A chained result is therefore only reproducible to
~1e-4 -- by ANY two fp32 implementations, the reference on another BLAS included -- while each iteration on identical
inputs is reproducible to rounding.  So the chain is stored link by link: x0, x1 = f(x0), x2 = f(x1), x3 = f(x2) with
f = one reference iteration (polyblur_deblurring(n_iter=3) IS f o f o f: deblurring.py:68-88), for both methods.

    python tests/golden/make_golden_kersize_chain.py      # writes tests/golden/pipeline_kersize_chain.npz
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

KW = dict(c=0.362, b=0.468, alpha=6, beta=1)
x, _ = synthetic_blurry_batch(1, 3, 72, 96, seed0=6161)
d = {"x0": x}
for k in (4, 12, 24, 36):
    for method in ("fft", "direct"):
        cur = x
        for it in range(3):
            cur = polyblur_deblurring(torch.from_numpy(cur.copy()), n_iter=1, ker_size=k, method=method, **KW).numpy()
            d["k%d_%s_x%d" % (k, method, it + 1)] = cur
        chained = polyblur_deblurring(torch.from_numpy(x.copy()), n_iter=3, ker_size=k, method=method, **KW).numpy()
        assert np.array_equal(chained, cur), (k, method)          # n_iter=3 is the three links
np.savez_compressed(os.path.join(HERE, "pipeline_kersize_chain.npz"), **d)
print({k: v.shape for k, v in d.items()})
