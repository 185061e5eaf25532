"""direct_separable end to end on a few synthetic 4K images (both arrangements of the x-t pass occur): ms per call"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch
kw = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1)
for seed in (1, 2, 3, 4, 5, 6):
    xn, true = synthetic_blurry_batch(1, 3, 2160, 3840, seed0=seed)
    x = torch.from_numpy(xn).cuda()
    res = {}
    for m in ("fft", "direct_separable"):
        for _ in range(2): o, infos = polyblur_deblurring(x, method=m, return_info=True, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): polyblur_deblurring(x, method=m, **kw)
        torch.cuda.synchronize(); res[m] = (time.perf_counter() - t0) * 200
    th = [round(float(np.rad2deg(i["theta"][0]))) for i in infos]
    sg = [(round(float(i["sigma"][0]), 2), round(float(i["rho"][0]), 2)) for i in infos]
    print("seed %d true %s  est theta %s sigma/rho %s: exact %.3f ms, separable approx %.3f ms" % (seed, [round(t, 1) for t in true[0]], th, sg, res["fft"], res["direct_separable"]))
