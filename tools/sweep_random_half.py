"""temporaries='fp16' (fp16 images, fp16 Horner temporaries: stated tolerance 8e-3 against the fp32 oracle on the fp16-rounded input,
same theta sequence as the fp32-temporaries call) on random shapes and options.  python tools/sweep_random_half.py [first last]"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch
from test_gpu_parity import _random_case
a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 100)
bad = 0; worst = 0.0; skipped = 0
for i in range(a, b):
    rng = np.random.default_rng(33000 + i)
    _, kw, coef = _random_case(11000 + i)
    kw.pop("edgetaping", None)                      # (not built for fp16 temporaries: raises PB_ERR_UNSUPPORTED by design)
    coef["alpha"], coef["beta"] = 6.0, 1.0          # (the tolerance is stated for the BASELINE coefficients: |a2| = 9)
    B, C = int(rng.integers(1, 4)), int(rng.choice([1, 3]))
    H, W = (int(rng.integers(40, 260)), int(rng.integers(40, 330))) if rng.random() < 0.6 else (int(rng.integers(260, 900)), int(rng.integers(330, 1300)))
    x16 = synthetic_blurry_batch(B, C, H, W, seed0=9100 + i)[0].astype(np.float16)
    xt = torch.from_numpy(x16).cuda()
    try:
        got, gi = polyblur_deblurring(xt, return_info=True, temporaries="fp16", **kw, **coef)
    except Exception as e:
        bad += 1; print("case", i, (B, C, H, W), kw, "raised", type(e).__name__, str(e)[:160]); continue
    base, bi = polyblur_deblurring(xt, return_info=True, **kw, **coef)
    want = ref.polyblur_deblurring(x16.astype(np.float32), **kw, **coef)
    same = all(np.array_equal(p["theta"], q["theta"]) for p, q in zip(gi, bi))
    err = float(np.abs(got.float().cpu().numpy() - want).max()); e0 = float(np.abs(base.float().cpu().numpy() - want).max())
    if e0 >= 2e-3: skipped += 1; continue            # (the fp32-temporaries call itself off the oracle: a flipped near-tie, the other sweeps' business)
    if not same or err >= 8e-3:
        bad += 1; print("case", i, (B, C, H, W), kw, "same theta as fp32 temporaries", same, "err %.3e (fp32 temporaries %.3e)" % (err, e0), flush=True)
    else: worst = max(worst, err)
print("fp16-temporaries cases %d..%d: %d outside 8e-3 / other theta / raised, %d skipped; worst inside %.3e" % (a, b, bad, skipped, worst))
