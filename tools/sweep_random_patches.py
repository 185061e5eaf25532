"""PolyblurDeblurring(patch_decomposition=True) on random shapes, patch sizes, overlaps and options against the oracle's restatement
(the reference's own branch raises NameError: unpinned), and grouped differently: python tools/sweep_random_patches.py [first last]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import polyblur_ref as ref
from polyblur_amd import PolyblurDeblurring
from polyblur_amd.synthetic import synthetic_blurry_batch
a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 60)
bad = 0; worst = 0.0
for i in range(a, b):
    rng = np.random.default_rng(47000 + i)
    B, C = int(rng.integers(1, 3)), int(rng.choice([1, 3]))
    H, W = int(rng.integers(60, 520)), int(rng.integers(60, 640))
    ps = int(rng.choice([64, 96, 128, 200, 256, 400])); ov = float(rng.choice([0.1, 0.25, 0.4, 0.5])); bs = int(rng.choice([1, 3, 8, 64]))
    kw = dict(n_iter=int(rng.integers(1, 4)), c=float(rng.uniform(0.3, 0.4)), b=float(rng.uniform(0.4, 0.8)), alpha=float(rng.choice([2, 6])), beta=float(rng.choice([1, 3])),
              remove_halo=bool(rng.integers(0, 2)), edgetaping=bool(rng.integers(0, 3) == 0), prefiltering=bool(rng.integers(0, 3) == 0), method=str(rng.choice(["fft", "direct"])))
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=5300 + i)
    from polyblur_amd.deblurring import patch_grid
    gg = patch_grid(H // 2 * 2, W // 2 * 2, (ps, ps), ov)
    if gg["n_i"] < 1 or gg["n_j"] < 1:                  # (the reference's lattice holds no patch: refused by the host, tests/test_capi_cpu.py)
        try:
            PolyblurDeblurring(patch_decomposition=True, patch_size=ps, patch_overlap=ov)(torch.from_numpy(x).cuda(), **kw)
            bad += 1; print("case", i, "an empty lattice was not refused")
        except ValueError:
            pass
        continue
    try:
        got = PolyblurDeblurring(patch_decomposition=True, patch_size=ps, patch_overlap=ov, batch_size=bs)(torch.from_numpy(x).cuda(), **kw)
        got1 = PolyblurDeblurring(patch_decomposition=True, patch_size=ps, patch_overlap=ov, batch_size=1)(torch.from_numpy(x).cuda(), **kw)
        want = ref.PolyblurDeblurring(patch_decomposition=True, patch_size=ps, patch_overlap=ov)(x, **kw)
        err = float(np.abs(got.cpu().numpy() - want).max()); same = bool(torch.equal(got, got1))
    except Exception as e:
        err = float("inf"); same = False; print("case", i, "raised", type(e).__name__, str(e)[:200])
    if err < 1e-4 and same: worst = max(worst, err)
    else:
        bad += 1; print("case", i, (B, C, H, W), ps, ov, bs, kw, "err %.3e" % err, "same bits grouped by 1:", same, flush=True)
print("patch cases %d..%d: %d outside 1e-4 or depending on the grouping; worst inside %.3e" % (a, b, bad, worst))
