"""Random sweep over the sizes between the small and the large sweeps, and over narrow / tall images (few tiles along one axis, many
along the other): every option, both boundary models, fp32 / fp16, against the oracle.  python tools/sweep_random_mid.py [first last]"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch
from test_gpu_parity import _random_case
a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 120)
bad = 0; worst = {np.float32: 0.0, np.float16: 0.0}
for i in range(a, b):
    rng = np.random.default_rng(52000 + i)
    _, kw, coef = _random_case(7000 + i)
    fam = int(rng.integers(0, 3))
    if fam == 0: H, W = int(rng.integers(150, 700)), int(rng.integers(200, 1000))
    elif fam == 1: H, W = int(rng.integers(200, 1500)), int(rng.integers(26, 110))
    else: H, W = int(rng.integers(26, 110)), int(rng.integers(200, 1500))
    B, C = int(rng.integers(1, 4)), int(rng.choice([1, 3]))
    dt = np.float16 if rng.integers(0, 4) == 0 else np.float32
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=3100 + i)
    x = x.astype(dt)
    got, infos = polyblur_deblurring(torch.from_numpy(x).cuda(), return_info=True, **kw, **coef)
    want, winfos = ref.polyblur_deblurring(x.astype(np.float32), return_info=True, **kw, **coef)
    same = all(np.array_equal(p["theta"], q["theta"]) for p, q in zip(infos, winfos))
    err = float(np.abs(got.float().cpu().numpy() - want).max())
    tol = 5e-5 if dt == np.float32 else 2e-3
    if same: worst[dt] = max(worst[dt], err)
    if not same or err >= tol:
        bad += 1
        print("case", i, (B, C, H, W), dt.__name__, kw, coef, "same_theta", same, "err %.3e" % err, flush=True)
print("mid cases %d..%d: %d outside tolerance, worst agreeing error fp32 %.3e, fp16 I/O %.3e" % (a, b, bad, worst[np.float32], worst[np.float16]))
