#!/usr/bin/env python3
"""Time of a whole 4K call by ker_size: the 25 x 25 record's bodies up to 25, the large-kernel pass (csrc/conv_big.hip) above."""
import sys
import time

import torch

sys.path.insert(0, ".")
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch

KW = dict(c=0.362, b=0.468, alpha=6, beta=1)
img, _ = synthetic_blurry_batch(1, 3, 2160, 3840, seed0=5)
d = torch.from_numpy(img).cuda()
for k in (25, 27, 31, 35, 41, 49):
    for method in ("fft", "direct"):
        polyblur_deblurring(d, n_iter=3, ker_size=k, method=method, **KW)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            polyblur_deblurring(d, n_iter=3, ker_size=k, method=method, **KW)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        print("ker_size %2d %-6s 4K n_iter=3: %8.3f ms  %7.1f MP/s" % (k, method, ms, 2160 * 3840 / ms / 1e3), flush=True)
