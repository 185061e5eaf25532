#!/usr/bin/env python3
"""Golden vectors from the REFERENCE itself for ker_size != 25 (this container only; same import recipe as
make_golden.py): the kernel support and, halved, the replicate pad both follow ker_size
(blur_estimation.py:211-232, utils.py:48-53), so the border behaviour of both methods changes with it.

    python tests/golden/make_golden_kersize.py      # writes tests/golden/pipeline_kersize.npz
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
x, _ = synthetic_blurry_batch(1, 3, 90, 122, seed0=5151)
d = {"x": x}
# odd sizes, and even ones: their grid arange(k) - (k - 1) // 2 is off-centre (blur_estimation.py:222), 'fft' rolls the kernel
# by k // 2 (filters.py:268-273) and F.conv2d's 'same' padding puts one row / column fewer in front than behind
# ... and sizes beyond the default 25 (the engine's large-kernel pass): odd, even, the largest built
for k in (5, 13, 21, 4, 12, 24, 31, 36, 49):
    for method in ("fft", "direct"):
        d["k%d_%s" % (k, method)] = polyblur_deblurring(torch.from_numpy(x.copy()), n_iter=2, ker_size=k, method=method, **KW).numpy()
d["k13_fft_taper_halo"] = polyblur_deblurring(torch.from_numpy(x.copy()), n_iter=2, ker_size=13, edgetaping=True,
                                             remove_halo=True, **KW).numpy()
np.savez_compressed(os.path.join(HERE, "pipeline_kersize.npz"), **d)
print({k: v.shape for k, v in d.items()})
