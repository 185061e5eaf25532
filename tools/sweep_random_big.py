"""Ad-hoc sweep at larger, odd sizes (many FFT factorisations, Bluestein lengths, interior + border tiles)."""
import sys, numpy as np, torch, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch
from test_gpu_parity import _random_case
a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 40)
bad = 0; worst = 0.0
for i in range(a, b):
    rng = np.random.default_rng(31000 + i)
    _, kw, coef = _random_case(2000 + i)
    B, C, H, W = 1, int(rng.choice([1, 3])), int(rng.integers(200, 900)), int(rng.integers(200, 900))
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=1500 + i)
    got, infos = polyblur_deblurring(torch.from_numpy(x).cuda(), return_info=True, **kw, **coef)
    want, winfos = ref.polyblur_deblurring(x, return_info=True, **kw, **coef)
    same = all(np.array_equal(p["theta"], q["theta"]) for p, q in zip(infos, winfos))
    err = float(np.abs(got.cpu().numpy() - want).max())
    worst = max(worst, err if same else 0.0)
    if not same or err >= 5e-5:
        bad += 1
        print("case", i, (B, C, H, W), kw, coef, "same_theta", same, "err %.3e" % err, flush=True)
print("cases %d..%d: %d outside tolerance, worst agreeing error %.3e" % (a, b, bad, worst))
