import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch
x, _ = synthetic_blurry_batch(2, 1, 141, 194, seed0=900 + 3 * 18)
coef = {'c': 0.34617493484855016, 'b': 0.6094157757422198, 'alpha': 6.0, 'beta': 4.0}
for k in (28, 24, 4):
    kw = dict(method="fft", ker_size=k, discard_saturation=True)
    x2 = polyblur_deblurring(torch.from_numpy(x).cuda(), n_iter=2, **kw, **coef).cpu().numpy()
    for support in ("full", "adaptive"):
        got, infos = polyblur_deblurring(torch.from_numpy(x2).cuda(), n_iter=1, return_info=True, support=support, **kw, **coef)
        want, winfos = ref.polyblur_deblurring(x2, n_iter=1, return_info=True, **kw, **coef)
        d = np.abs(got.cpu().numpy() - want)
        i = infos[0]
        print(k, support, "third iteration alone: max %.3e frac>2e-5 %.2e" % (d.max(), (d > 2e-5).mean()), float(i["sigma"][0]), float(i["rho"][0]), float(i["theta"][0]),
              "radius", int(i["radius"][0]), "sep", int(i["separable"][0]), "nphase", i["nphase"][0])
        kern = i["kernel"][0]
        wk = ref.gaussian_kernel_2d(winfos[0]["theta"], winfos[0]["sigma"], winfos[0]["rho"], k)[0]
        nz = np.argwhere(kern > 1e-12)
        print("   engine record taps > 1e-12 at (row, col) range", nz.min(0), nz.max(0), "sum", kern.sum(), " oracle kernel argmax", np.unravel_index(wk.argmax(), wk.shape), "centre val", wk.max())
        # where do they differ
        idx = np.unravel_index(d.argmax(), d.shape)
        print("   worst at", idx, got.cpu().numpy()[idx], want[idx])
