"""End-to-end ms per call of the headline image under each method / support policy: python tools/bench_methods.py [H W]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch, DEFAULT_SEED
H, W = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (2160, 3840)
x = torch.from_numpy(synthetic_blurry_batch(1, 3, H, W, seed0=DEFAULT_SEED)[0]).cuda()
kw = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1)
for name, extra in (("fft full", {}), ("fft adaptive", dict(support="adaptive")), ("direct_separable", dict(method="direct_separable")),
                    ("direct_separable adaptive", dict(method="direct_separable", support="adaptive"))):
    try:
        for _ in range(3): polyblur_deblurring(x, **kw, **extra)
    except Exception as e:
        print(name, "n/a", e); continue
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): polyblur_deblurring(x, **kw, **extra)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    print("%-28s %.3f ms  %8.0f MP/s" % (name, ms, H * W / 1e3 / ms))
if "--f16" in sys.argv:
    for B, Hh, Ww, it in ((1, 4320, 7680, 5), (16, 1080, 1920, 3)):
        nd = min(B, 4)
        xs = synthetic_blurry_batch(nd, 3, Hh, Ww, seed0=DEFAULT_SEED)[0]
        xh = torch.from_numpy(np.concatenate([xs] * (B // nd))).cuda().half()
        kk = dict(kw, n_iter=it)
        outs = {}
        for th_name, th in (("oblique as estimated", {}),):
            for tmp in ("fp32", "fp16"):
                for sup in ("full", "adaptive"):
                    for _ in range(2): o = polyblur_deblurring(xh, temporaries=tmp, support=sup, **kk)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(5): o = polyblur_deblurring(xh, temporaries=tmp, support=sup, **kk)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) * 200
                    outs[(tmp, sup)] = o
                    print("B=%d %dx%d n_iter=%d fp16 I/O, temporaries=%s support=%s: %.3f ms  %8.0f MP/s" % (B, Ww, Hh, it, tmp, sup, ms, B * Hh * Ww / 1e3 / ms))
        print("   max |fp16 temporaries - fp32 temporaries| =", float((outs[("fp16", "full")].float() - outs[("fp32", "full")].float()).abs().max()))
