#!/usr/bin/env python3
"""Through-memory spectral derivative (sides whose lines do not fit LDS) against the in-LDS one on a neighbouring size:
gradients checked against the oracle's numpy transform, then the time of a whole call.  GPU box only."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch

eng = get_engine(0)
KW = dict(c=0.362, b=0.468, alpha=6, beta=1)
for H, W in ((9000, 12000), (9001, 12001), (4000, 24000), (4000, 23000)):
    rng = np.random.default_rng(3)
    x = rng.random((1, 1, H, W), dtype=np.float32)
    gx, gy = eng.fourier_gradients(x)
    rx, ry = ref.spectral_gradients(x)
    ex = float(np.abs(np.asarray(gx) - rx).max()); ey = float(np.abs(np.asarray(gy) - ry).max())
    del gx, gy, rx, ry
    img, _ = synthetic_blurry_batch(1, 3, 64, 64, seed0=1)
    img = np.tile(img, (1, 1, H // 64 + 1, W // 64 + 1))[:, :, :H, :W].copy()
    d = torch.from_numpy(img).cuda()
    out = polyblur_deblurring(d, n_iter=3, **KW)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = polyblur_deblurring(d, n_iter=3, **KW)
    torch.cuda.synchronize()
    tp = (time.perf_counter() - t0) / 3
    print("%5d x %5d  gradients: max err %.2e / %.2e   whole call n_iter=3: %.2f ms = %.0f MP/s, finite %s" % (
        H, W, ex, ey, tp * 1e3, H * W / tp / 1e6, bool(torch.isfinite(out).all())), flush=True)
    del d, out
    torch.cuda.empty_cache()
