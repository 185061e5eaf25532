"""Ad-hoc sweep of the narrow I/O types against the oracle: python tools/sweep_random_io.py [first last]"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring, polyblur_deblurring_uint8
from polyblur_amd.synthetic import synthetic_blurry_batch
from test_gpu_parity import _random_case
a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 80)
bad = 0; w16 = 0.0; w8 = 0.0
for i in range(a, b):
    (B, C, H, W), kw, coef = _random_case(1000 + i)
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=900 + 3 * i)
    xh = x.astype(np.float16)
    got = polyblur_deblurring(torch.from_numpy(xh).cuda(), **kw, **coef).float().cpu().numpy()
    want, winfos = ref.polyblur_deblurring(xh.astype(np.float32), return_info=True, **kw, **coef)
    e16 = float(np.abs(got - want).max())
    u = ref.img_as_ubyte_from_float(x)
    gu = polyblur_deblurring_uint8(torch.from_numpy(u).cuda(), **kw, **coef).cpu().numpy()
    wu = np.stack([np.moveaxis(ref.polyblur_deblurring_uint8(np.moveaxis(u[j], 0, -1) if C == 3 else u[j, 0], **kw, **coef), -1, 0) if C == 3
                   else ref.polyblur_deblurring_uint8(u[j, 0], **kw, **coef)[None] for j in range(B)])
    d = np.abs(gu.astype(int) - wu.astype(int))
    e8 = int(d.max()); f8 = float(np.mean(d != 0))
    w16 = max(w16, e16); w8 = max(w8, f8)
    if e16 >= 1.5e-3 or e8 > 1 or f8 > 5e-3:
        bad += 1
        print("case", i, (B, C, H, W), kw, coef, "fp16 err %.3e" % e16, "u8 max", e8, "frac %.2e" % f8)
print("cases %d..%d: %d outside tolerance; worst fp16 error %.3e, worst u8 mismatch fraction %.2e" % (a, b, bad, w16, w8))
