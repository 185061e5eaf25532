"""Random sweep over the shapes whose line transforms take the one-plan kernels of csrc/lines_fixed.hip (both sides among the
compiled line lengths): random batch, channels, options, dtype and coefficients, each case against the oracle.
    python tools/sweep_fixed_plans.py [first last]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch

SIDES = [512, 640, 720, 768, 800, 960, 1024, 1080, 1200, 1280, 1440, 1536, 1600, 1920, 2048]


def case(i):
    rng = np.random.default_rng(88000 + i)
    while True:
        H, W = int(rng.choice(SIDES)), int(rng.choice(SIDES))
        if H * W <= 1.7e6:
            break
    B, C = int(rng.integers(1, 3)), int(rng.choice([1, 3]))
    kw = dict(n_iter=int(rng.integers(1, 4)), method=str(rng.choice(["fft", "direct"])),
              remove_halo=bool(rng.integers(0, 2)), edgetaping=bool(rng.integers(0, 4) == 0),
              prefiltering=bool(rng.integers(0, 3) == 0), discard_saturation=bool(rng.integers(0, 2)),
              q=float(rng.choice([0.0, 0.0, 0.0, 1e-3])))
    if kw["prefiltering"]:
        kw["prefilter"] = str(rng.choice(["bilateral", "domain_transform"]))
    coef = dict(c=float(rng.uniform(0.3, 0.4)), b=float(rng.uniform(0.4, 0.8)), alpha=float(rng.choice([2, 4, 6])),
                beta=float(rng.choice([1, 3, 4])))
    half = bool(rng.integers(0, 3) == 0)
    return (B, C, H, W), kw, coef, half


a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 40)
bad = 0; worst = 0.0; worst16 = 0.0
for i in range(a, b):
    (B, C, H, W), kw, coef, half = case(i)
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=1200 + 5 * i)
    xin = x.astype(np.float16) if half else x
    got, infos = polyblur_deblurring(torch.from_numpy(xin).cuda(), return_info=True, **kw, **coef)
    want, winfos = ref.polyblur_deblurring(xin.astype(np.float32), return_info=True, **kw, **coef)
    same = all(np.array_equal(p["theta"], q["theta"]) for p, q in zip(infos, winfos))
    err = float(np.abs(got.float().cpu().numpy() - want).max())
    tol = 1e-3 if half else 5e-5
    if same:
        if half: worst16 = max(worst16, err)
        else: worst = max(worst, err)
    if not same or err >= tol:
        bad += 1
        print("case", i, (B, C, H, W), "fp16" if half else "fp32", kw, coef, "same_theta", same, "err %.3e" % err, flush=True)
print("fixed-plan cases %d..%d: %d outside tolerance, worst agreeing error fp32 %.3e, fp16 I/O %.3e" % (a, b, bad, worst, worst16))
