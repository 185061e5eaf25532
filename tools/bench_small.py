"""Latency of whole calls on small images, both dense bodies: python tools/bench_small.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from polyblur_amd import polyblur_deblurring
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch
KW = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1)
eng = get_engine(0)
for (h, w) in ((500, 700), (512, 512), (1080, 1920)):
    x = torch.from_numpy(synthetic_blurry_batch(1, 3, h, w, seed0=7)[0]).cuda()
    for mode in ("auto", "stencil"):
        eng.set_dense_eval(mode, 16) if mode == "auto" else eng.set_dense_eval("stencil")
        for _ in range(5): polyblur_deblurring(x, **KW)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): polyblur_deblurring(x, **KW)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
        print("%dx%d %-8s %.3f ms/call  %.0f MP/s" % (w, h, mode, ms, h * w / 1e3 / ms))
