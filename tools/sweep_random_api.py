"""Random sweep over the parts of the API the other sweeps hold fixed: n_angles, n_interpolated_angles, multichannel_kernel, the
adaptive support policy, sigma_s / sigma_r, HWC ndarray / 8-bit / fp16 inputs, PolyblurDeblurring with and without patch
decomposition -- each case against the oracle.  python tools/sweep_random_api.py [first last]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring, PolyblurDeblurring
from polyblur_amd.synthetic import synthetic_blurry_batch
a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 150)
bad = 0; worst = 0.0
for i in range(a, b):
    rng = np.random.default_rng(64000 + i)
    B, C = int(rng.integers(1, 4)), int(rng.choice([1, 3]))
    H, W = int(rng.integers(60, 420)), int(rng.integers(60, 520))
    kw = dict(n_iter=int(rng.integers(1, 4)), method=str(rng.choice(["fft", "direct"])), n_angles=int(rng.choice([3, 4, 6, 8, 12])),
              n_interpolated_angles=int(rng.choice([12, 24, 30, 45, 60])), multichannel_kernel=bool(rng.integers(0, 2)),
              remove_halo=bool(rng.integers(0, 2)), edgetaping=bool(rng.integers(0, 3) == 0), prefiltering=bool(rng.integers(0, 3) == 0),
              discard_saturation=bool(rng.integers(0, 2)), q=float(rng.choice([0.0, 0.0, 1e-3])),
              sigma_s=float(rng.uniform(1.0, 6.0)), sigma_r=float(rng.uniform(0.2, 1.0)),
              c=float(rng.uniform(0.3, 0.4)), b=float(rng.uniform(0.4, 0.8)), alpha=float(rng.choice([2, 4, 6])), beta=float(rng.choice([1, 3, 4])))
    if kw["prefiltering"]: kw["prefilter"] = str(rng.choice(["bilateral", "domain_transform"]))
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=4100 + i)
    mode = int(rng.integers(0, 4))
    try:
        if mode == 0:                                  # (B,C,H,W) cuda tensor
            got = polyblur_deblurring(torch.from_numpy(x).cuda(), **kw).cpu().numpy(); want = ref.polyblur_deblurring(x, **kw); tol = 5e-5
        elif mode == 1:                                # HWC / HW ndarray
            im = np.ascontiguousarray(np.moveaxis(x[0], 0, -1)) if C == 3 else x[0, 0]
            got = polyblur_deblurring(im, **kw); want = ref.polyblur_deblurring(im, **kw); tol = 5e-5
        elif mode == 2:                                # the module, whole images
            kwm = {k: v for k, v in kw.items() if k not in ("prefilter",)}
            got = PolyblurDeblurring()(torch.from_numpy(x).cuda(), **kwm).cpu().numpy(); want = ref.polyblur_deblurring(x, **kwm); tol = 5e-5
        else:                                          # fp16 tensor
            xh = x.astype(np.float16)
            got = polyblur_deblurring(torch.from_numpy(xh).cuda(), **kw).float().cpu().numpy(); want = ref.polyblur_deblurring(xh.astype(np.float32), **kw); tol = 2e-3
        err = float(np.abs(np.asarray(got, np.float32) - want).max())
    except Exception as e:
        err = float("inf"); print("case", i, "raised", type(e).__name__, str(e)[:200])
    # (a flipped near-tie in a direction estimate shows as a large error: reported, then looked at)
    if err < tol: worst = max(worst, err if tol < 1e-3 else 0.0)
    else:
        bad += 1
        print("case", i, (B, C, H, W), "mode", mode, kw, "err %.3e" % err, flush=True)
print("api cases %d..%d: %d outside tolerance, worst fp32 error inside %.3e" % (a, b, bad, worst))
