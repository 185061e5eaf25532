import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch
KW = dict(c=0.362, b=0.468, alpha=6, beta=1)
eng = get_engine(0)
for seed in (41, 42, 43):
    x, _ = synthetic_blurry_batch(2, 3, 96, 140, seed0=seed)
    kw = dict(n_iter=2, prefiltering=True, sigma_s=2.0, sigma_r=0.8, **KW)
    want = ref.polyblur_deblurring(x, prefilter="normalized_convolution", **kw)
    for mode in ("auto", "stencil"):
        eng.set_dense_eval(mode, 36) if mode == "auto" else eng.set_dense_eval("stencil")
        got = polyblur_deblurring(torch.from_numpy(x), prefilter="normalized_convolution", **kw).numpy()
        d = np.abs(got - want)
        print(seed, mode, "frac>2e-5 %.2e  n %d  max %.2e" % (np.mean(d > 2e-5), np.sum(d > 2e-5), d.max()))
