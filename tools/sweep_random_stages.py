"""The stage entry points on random shapes against the oracle: fourier_gradients (sides with large prime factors, sides beyond the
in-LDS transform), halo_mask, the normalized-convolution filter, the bilateral filter, and method='direct_separable' against the
oracle's x-t restatement.  python tools/sweep_random_stages.py [first last]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch
a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 120)
eng = get_engine(0)
bad = 0; worst = {}
def note(name, err, tol, ctx):
    global bad
    worst[name] = max(worst.get(name, 0.0), err if err < tol else 0.0)
    if not err < tol:
        bad += 1; print(name, ctx, "err %.3e (tol %.1e)" % (err, tol), flush=True)
for i in range(a, b):
    rng = np.random.default_rng(91000 + i)
    B, C = int(rng.integers(1, 4)), int(rng.choice([1, 3]))
    H, W = int(rng.integers(4, 400)), int(rng.integers(4, 520))
    if rng.random() < 0.15: W = int(rng.choice([8209, 9001, 8192, 12289]))     # beyond the in-LDS transform / the largest in it
    if rng.random() < 0.15: H = int(rng.choice([1009, 2053, 4099, 8191]))       # primes: the chirp-z path
    if H * W > 3_000_000: H = max(4, 3_000_000 // W)
    x = rng.random((B, C, H, W), dtype=np.float32)
    gx, gy = eng.fourier_gradients(x)
    rx, ry = ref.spectral_gradients(x)
    sc = max(1.0, float(np.abs(rx).max()), float(np.abs(ry).max()))
    note("fourier_gradients", max(float(np.abs(gx - rx).max()), float(np.abs(gy - ry).max())) / sc, 8e-6, (B, C, H, W))
    if H >= 8 and W >= 8 and H * W < 400_000:
        y = np.clip(x + 0.05 * rng.standard_normal(x.shape).astype(np.float32), 0, 1)
        note("halo_mask", float(np.abs(eng.halo_mask(x, y, rx, ry) - ref.halo_masking(x, y, (rx, ry))).max()), 2e-6, (B, C, H, W))
        note("bilateral", float(np.abs(eng.bilateral5(x) - ref.bilateral_filter(x)).max()), 3e-6, (B, C, H, W))
        ss, sr, N = float(rng.uniform(2, 40)), float(rng.uniform(0.2, 1.0)), int(rng.integers(1, 4))
        note("normalized_convolution", float(np.abs(eng.dt_normalized_convolution(x, ss, sr, N) - ref.normalized_convolution(x, ss, sr, N)).max()), 1e-5, (B, C, H, W, ss, sr, N))
    if 60 <= H <= 300 and 60 <= W <= 400 and i % 3 == 0:
        xs, _ = synthetic_blurry_batch(B, C, H, W, seed0=8100 + i)
        kw = dict(n_iter=int(rng.integers(1, 4)), c=0.362, b=0.468, alpha=6.0, beta=1.0, method="direct_separable")
        got, gi = polyblur_deblurring(torch.from_numpy(xs).cuda(), return_info=True, **kw)
        want, wi = ref.polyblur_deblurring(xs, return_info=True, **kw)
        same = all(np.array_equal(p["theta"], q["theta"]) for p, q in zip(gi, wi))
        note("direct_separable", float(np.abs(got.cpu().numpy() - want).max()) if same else 1.0, 1e-4, (B, C, H, W, kw["n_iter"]))
print("stage cases %d..%d: %d outside tolerance; worst inside:" % (a, b, bad), {k: "%.2e" % v for k, v in worst.items()})
