"""NC-prefilter pipeline, iteration by iteration on IDENTICAL inputs (the engine's output of iteration k feeds both
sides' iteration k + 1): how many samples differ, for both dense bodies and several seeds."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring, _capi as capi
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch
KW = dict(c=0.362, b=0.468, alpha=6, beta=1)
eng = get_engine(0)
for seed in (41, 43, 47, 53):
    x, _ = synthetic_blurry_batch(2, 3, 96, 140, seed0=seed)
    kw = dict(n_iter=1, prefiltering=True, sigma_s=2.0, sigma_r=0.8, **KW)
    for body in ("auto", "stencil"):
        eng.set_dense_eval(body, 0)
        cur = x
        for k in range(3):
            got = polyblur_deblurring(torch.from_numpy(cur), prefilter="normalized_convolution", **kw).numpy()
            want = ref.polyblur_deblurring(cur, prefilter="normalized_convolution", **kw)
            d = np.abs(got - want)
            print("seed %d body %-7s iter %d: max %.3g, above 2e-5: %d of %d, above 1e-4: %d" % (seed, body, k + 1, d.max(), (d > 2e-5).sum(), d.size, (d > 1e-4).sum()))
            cur = got
eng.set_dense_eval("auto", capi.PB_DENSE_MIN_PHASES)
