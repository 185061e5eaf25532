"""Debug: the pipeline / inverse filter through the tile-spectrum body against the same call through the stencil body."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import _capi as capi, polyblur_deblurring
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch
eng = get_engine(0)
KW = dict(c=0.362, b=0.468, alpha=6, beta=1)
def report(tag, a, b):
    d = np.abs(a - b); bad = d > 1e-4
    print(tag, "max %.3g nbad %d of %d" % (d.max(), bad.sum(), bad.size))
    if bad.any():
        for pl in range(bad.shape[0] * bad.shape[1]):
            m = bad[pl // bad.shape[1], pl % bad.shape[1]]
            if not m.any(): continue
            ys, xs = np.nonzero(m)
            print("  plane", pl, "nbad", m.sum(), "rows %d..%d cols %d..%d" % (ys.min(), ys.max(), xs.min(), xs.max()),
                  "rows%40", np.unique(ys % 40).tolist()[:40], "ncols%80", len(np.unique(xs % 80)))
            print("   distinct rows", len(np.unique(ys)), "distinct cols", len(np.unique(xs)), "first rows", np.unique(ys)[:10].tolist(), "first cols", np.unique(xs)[:10].tolist())
sizes = [(1, 2160, 3840), (1, 1080, 1920), (1, 500, 700)]
for (B, H, W) in sizes:
    x, _ = synthetic_blurry_batch(B, 3, H, W, seed0=20260929)
    xt = torch.from_numpy(x).cuda()
    for n_iter in (1, 3):
        eng.set_dense_eval("auto", 16); a = polyblur_deblurring(xt, n_iter=n_iter, **KW).cpu().numpy()
        eng.set_dense_eval("stencil", 0); b = polyblur_deblurring(xt, n_iter=n_iter, **KW).cpu().numpy()
        report("pipeline %dx%d n_iter=%d" % (H, W, n_iter), a, b)
    th = np.deg2rad(np.float32(66.0))
    for bnd in (capi.PB_WRAP, capi.PB_ZERO):
        eng.set_dense_eval("auto", 16)
        buf = eng.make_kernels([2.0] * B, [1.3] * B, [th] * B, support=0)
        a = eng.inverse_filter(x, buf, 6.0, 1.0, bnd)
        eng.set_dense_eval("stencil", 0)
        buf = eng.make_kernels([2.0] * B, [1.3] * B, [th] * B, support=0)
        b = eng.inverse_filter(x, buf, 6.0, 1.0, bnd)
        report("inverse_filter %dx%d boundary %d" % (H, W, bnd), a, b)
