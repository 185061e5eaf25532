"""An image's result must not depend on the batch it travels in: random shapes and options, a batch against its images one at a
time, bit for bit (outputs and records).  python tools/sweep_random_batch.py [first last]"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch
from test_gpu_parity import _random_case
a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 100)
bad = 0
for i in range(a, b):
    rng = np.random.default_rng(81000 + i)
    _, kw, coef = _random_case(9000 + i)
    B, C = int(rng.integers(2, 9)), int(rng.choice([1, 3]))
    fam = int(rng.integers(0, 3))
    H, W = (int(rng.integers(30, 200)), int(rng.integers(30, 260))) if fam == 0 else ((int(rng.integers(200, 800)), int(rng.integers(200, 1100))) if fam == 1 else (int(rng.choice([360, 540, 720, 1080])), int(rng.choice([640, 960, 1280, 1920]))))
    dt = torch.float16 if rng.integers(0, 4) == 0 else torch.float32
    x = torch.from_numpy(synthetic_blurry_batch(B, C, H, W, seed0=6100 + 11 * i)[0]).cuda().to(dt)
    full, infos = polyblur_deblurring(x, return_info=True, **kw, **coef)
    for j in sorted(set([0, B - 1, int(rng.integers(0, B))])):
        one, oinfos = polyblur_deblurring(x[j:j + 1], return_info=True, **kw, **coef)
        same = torch.equal(full[j:j + 1], one) and all(np.array_equal(np.asarray(p[f])[j], np.asarray(q[f])[0]) for p, q in zip(infos, oinfos) for f in ("theta", "sigma", "rho", "mags"))
        if not same:
            bad += 1
            print("case", i, (B, C, H, W), str(dt), "image", j, kw, "max diff %.3e" % float((full[j:j + 1].float() - one.float()).abs().max()), flush=True)
            break
print("batch cases %d..%d: %d with an image that differs between the batch and the lone call" % (a, b, bad))
