"""edgetaping=True on random small and narrow shapes against the oracle: python tools/sweep_random_taper.py [first last]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch
a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 150)
bad = 0; worst = 0.0
for i in range(a, b):
    rng = np.random.default_rng(31000 + i)
    B, C = int(rng.integers(1, 4)), int(rng.choice([1, 3]))
    H = int(rng.choice([rng.integers(26, 120), rng.integers(120, 420)])); W = int(rng.choice([rng.integers(26, 120), rng.integers(120, 420)]))
    kw = dict(n_iter=int(rng.integers(1, 4)), edgetaping=True, method=str(rng.choice(["fft", "direct"])), remove_halo=bool(rng.integers(0, 2)),
              c=float(rng.uniform(0.3, 0.4)), b=float(rng.uniform(0.4, 0.8)), alpha=float(rng.choice([2.0, 6.0])), beta=float(rng.choice([1.0, 3.0])))
    if rng.random() < 0.3: kw["ker_size"] = int(rng.choice([11, 17, 25, 31]))
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=900 + 3 * i)
    got, infos = polyblur_deblurring(torch.from_numpy(x).cuda(), return_info=True, **kw)
    want, winfos = ref.polyblur_deblurring(x, return_info=True, **kw)
    same = all(np.array_equal(p["theta"], q["theta"]) for p, q in zip(infos, winfos))
    err = float(np.abs(got.cpu().numpy() - want).max())
    worst = max(worst, err if same else 0.0)
    if not same or err >= 5e-5:
        bad += 1
        print("case", i, (B, C, H, W), kw, "same_theta", same, "err %.3e" % err)
print("taper cases %d..%d: %d outside tolerance, worst agreeing error %.3e" % (a, b, bad, worst))
