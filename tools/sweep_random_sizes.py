"""Random sweep over what round 3 added to the accepted inputs: ker_size 2..49 (even sizes off-centre; above 25 the
large-kernel pass), image sides whose lines do not fit LDS, the adaptive support policy -- each case against the oracle.
    python tools/sweep_random_sizes.py [first last]         (PB_POLY1=1 in the environment adds the one-pass polynomial)"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch


def case(i):
    rng = np.random.default_rng(77000 + i)
    B, C = int(rng.integers(1, 3)), int(rng.choice([1, 3]))
    k = int(rng.choice([int(rng.integers(2, 50)), 25, int(rng.integers(26, 50))]))
    H, W = int(rng.integers(2 * k + 2, 200)), int(rng.integers(2 * k + 2, 260))
    if rng.integers(0, 4) == 0:                      # a side beyond the in-LDS transform, the other kept short
        long_side = int(rng.choice([8200, 8209, 9001, 20500 + int(rng.integers(0, 500))]))
        short = int(rng.integers(2 * k + 2, 2 * k + 40))
        H, W = (short, long_side) if rng.integers(0, 2) else (long_side, short)
        B = 1
    plain = False                                    # edgetaping takes every size since round 6 (even ones; above 25 through conv_big.hip)
    kw = dict(n_iter=int(rng.integers(1, 4)), method=str(rng.choice(["fft", "direct"])), ker_size=k,
              remove_halo=bool(rng.integers(0, 2)), edgetaping=(not plain) and bool(rng.integers(0, 2)),
              prefiltering=bool(rng.integers(0, 3) == 0), discard_saturation=bool(rng.integers(0, 2)),
              q=float(rng.choice([0.0, 0.0, 1e-3])))
    if kw["prefiltering"]:
        kw["prefilter"] = str(rng.choice(["bilateral", "domain_transform"]))
    coef = dict(c=float(rng.uniform(0.3, 0.4)), b=float(rng.uniform(0.4, 0.8)), alpha=float(rng.choice([2, 4, 6])),
                beta=float(rng.choice([1, 3, 4])))
    support = str(rng.choice(["full", "adaptive"]))
    return (B, C, H, W), kw, coef, support


a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 120)
bad = 0; worst = 0.0
for i in range(a, b):
    (B, C, H, W), kw, coef, support = case(i)
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=900 + 3 * i)
    got, infos = polyblur_deblurring(torch.from_numpy(x).cuda(), return_info=True, support=support, **kw, **coef)
    want, winfos = ref.polyblur_deblurring(x, return_info=True, **kw, **coef)
    same = all(np.array_equal(p["theta"], q["theta"]) for p, q in zip(infos, winfos))
    err = float(np.abs(got.cpu().numpy() - want).max())
    # An even ker_size under 'fft' is the Gaussian centred on offset +1 (filters.py:268-273): its transform carries the shift's
    # phase, so a3 K^3 + a2 K^2 + a1 K + b reaches |a3| + |a2| + |a1| + |b| near Nyquist (6.3 for alpha 6, beta 4 against 1.7
    # for the centred kernel) and every later iteration multiplies the earlier ones' rounding differences by that -- each
    # iteration alone agrees to 1e-6 (tools/dbg_case18.py).  Those chains get a tolerance that admits the amplification.
    tol = 3e-4 if (kw["ker_size"] % 2 == 0 and kw["method"] == "fft" and kw["n_iter"] > 1) else 5e-5
    if same and tol == 5e-5: worst = max(worst, err)
    if not same or err >= tol:
        bad += 1
        print("case", i, (B, C, H, W), support, kw, coef, "same_theta", same, "err %.3e" % err, flush=True)
print("cases %d..%d: %d outside tolerance, worst agreeing error %.3e" % (a, b, bad, worst))
