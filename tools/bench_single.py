"""Whole calls on ONE image of common sizes, plain and with each optional stage: python tools/bench_single.py"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch
KW = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1)
OPTS = {"plain": {}, "remove_halo": dict(remove_halo=True), "prefilter bilateral": dict(prefiltering=True),
        "prefilter domain transform": dict(prefiltering=True, prefilter="domain_transform"), "edgetaping": dict(edgetaping=True),
        "direct": dict(method="direct"), "halo + bilateral + edgetaper": dict(remove_halo=True, prefiltering=True, edgetaping=True)}
for (h, w) in ((500, 700), (720, 1280), (1080, 1920), (2160, 3840)):
    x = torch.from_numpy(synthetic_blurry_batch(1, 3, h, w, seed0=7)[0]).cuda()
    row = []
    for name, o in OPTS.items():
        kw = dict(KW, **o)
        for _ in range(5): polyblur_deblurring(x, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): polyblur_deblurring(x, **kw)
        torch.cuda.synchronize(); row.append("%s %.3f" % (name, (time.perf_counter() - t0) / 30 * 1e3))
    print("%dx%d ms per call: " % (w, h) + "; ".join(row))
