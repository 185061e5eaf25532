"""The inverse filter, convolve2d and the edgetaper with CALLER-SUPPLIED 25 x 25 kernels of every kind (Gaussians, asymmetric blobs, motion
lines, boxes, shifted deltas, sparse taps; a batch mixes them) on random shapes, coefficients and boundary models, against the oracle.
python tools/sweep_random_kernels.py [first last]"""
import sys, numpy as np
sys.path.insert(0, '.')
from oracle import polyblur_ref as ref
from polyblur_amd import _capi as capi
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch
K = capi.PB_KSIZE


def kernel(rng):
    k = np.zeros((K, K), np.float64); c = K // 2
    kind = int(rng.integers(0, 7))
    yy, xx = np.mgrid[-c:c + 1, -c:c + 1].astype(np.float64)
    if kind == 0:                                   # an oblique Gaussian
        s, r, t = rng.uniform(0.3, 4.0), rng.uniform(0.3, 4.0), rng.uniform(0, np.pi)
        u, v = np.cos(t) * xx + np.sin(t) * yy, -np.sin(t) * xx + np.cos(t) * yy
        k = np.exp(-0.5 * (u * u / (s * s) + v * v / (r * r)))
    elif kind == 1:                                 # an asymmetric blob: a Gaussian off the centre, times a ramp
        s = rng.uniform(0.8, 3.0); ox, oy = rng.uniform(-3, 3, 2)
        k = np.exp(-0.5 * ((xx - ox) ** 2 + (yy - oy) ** 2) / (s * s)) * (1.0 + 0.05 * xx + 0.03 * yy).clip(0.1)
    elif kind == 2:                                 # a motion line
        n = int(rng.integers(2, 12)); t = rng.uniform(0, np.pi)
        for a in np.linspace(-n, n, 8 * n + 1):
            k[int(round(c + a * np.sin(t))), int(round(c + a * np.cos(t)))] += 1.0
    elif kind == 3:                                 # a box
        h, w = int(rng.integers(0, 6)), int(rng.integers(0, 6)); k[c - h:c + h + 1, c - w:c + w + 1] = 1.0
    elif kind == 4:                                 # a shifted delta plus a little of its neighbour
        oy, ox = (int(v) for v in rng.integers(-4, 5, 2)); k[c + oy, c + ox] = 0.8; k[c, c] += 0.2
    elif kind == 5:                                 # sparse random taps
        for _ in range(int(rng.integers(2, 9))):
            k[int(rng.integers(c - 6, c + 7)), int(rng.integers(c - 6, c + 7))] += rng.uniform(0.1, 1.0)
    else:                                           # a rank-1 (separable) kernel that is not Gaussian
        a = np.exp(-np.abs(np.arange(-c, c + 1)) / rng.uniform(0.5, 3.0)); b = (np.abs(np.arange(-c, c + 1)) <= int(rng.integers(0, 5))).astype(np.float64)
        k = np.outer(a, b)
    return (k / k.sum()).astype(np.float32)


a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 150)
eng = get_engine(0)
bad = 0; worst = 0.0
for i in range(a, b):
    rng = np.random.default_rng(73000 + i)
    B, C = int(rng.integers(1, 5)), int(rng.choice([1, 3]))
    H, W = (int(rng.integers(30, 260)), int(rng.integers(30, 330))) if rng.random() < 0.7 else (int(rng.integers(260, 900)), int(rng.integers(330, 1300)))
    alpha, beta = float(rng.choice([2.0, 4.0, 6.0])), float(rng.choice([1.0, 3.0, 4.0]))
    method = str(rng.choice(["fft", "direct"])); taper = bool(rng.integers(0, 3) == 0)
    boundary = capi.PB_WRAP if method == "fft" else capi.PB_ZERO
    ks = np.stack([kernel(rng) for _ in range(B)])
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=7100 + i)
    buf = eng.set_kernels(ks)
    res = {}
    got = eng.inverse_filter(x, buf, alpha, beta, boundary, edgetaping=taper)
    res["inverse"] = float(np.abs(got - ref.inverse_filtering_rank3(x, ks[:, None], alpha, beta, do_edgetaper=taper, method=method)).max())
    xp = ref.replicate_pad(x, K // 2)
    res["convolve2d"] = float(np.abs(eng.convolve2d(xp, buf, boundary) - ref.convolve2d(xp, ks[:, None], method=method)).max())
    if min(H, W) >= 2:
        res["edgetaper"] = float(np.abs(eng.edgetaper(xp, buf, boundary) - ref.edgetaper(xp, ks[:, None], method=method)).max())
    # the polynomial amplifies by up to |a3| + |a2| + |a1| + b: tolerances scale with it
    amp = abs(alpha / 2 - beta + 2) + abs(3 * beta - alpha - 6) + abs(5 - 3 * beta + alpha / 2) + beta
    tol = dict(inverse=2e-6 * amp, convolve2d=2e-6, edgetaper=4e-6)
    if any(res[k] >= tol[k] for k in res):
        bad += 1; print("case", i, (B, C, H, W), method, "taper" if taper else "", alpha, beta, {k: "%.2e" % v for k, v in res.items()}, "selection", eng.body_selection(B).tolist(), flush=True)
    else: worst = max(worst, res["inverse"] / amp)
print("kernel cases %d..%d: %d outside tolerance; worst inverse-filter error / amplification inside %.3e" % (a, b, bad, worst))
