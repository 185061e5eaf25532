"""ONE engine, a random SEQUENCE of different calls on different shapes (whole calls with options, caller kernels through the stage
entry points, Gaussian records, the filters, the gradients): every result against the oracle -- what a context remembers between
calls (records' facts, spectra, selections, scratch, reflected taps) must never leak into the next.  python tools/sweep_random_sequence.py [steps seed [mid]]"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
from oracle import polyblur_ref as ref
from polyblur_amd import polyblur_deblurring, _capi as capi
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch
from test_gpu_parity import _random_case
exec(open('tools/sweep_random_kernels.py').read().split("a, b = (int(v)")[0].split('"""', 2)[2])      # kernel(rng)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MID = len(sys.argv) > 3 and sys.argv[3] == "mid"
eng = get_engine(0)
K = capi.PB_KSIZE
bad = 0; count = {}
for t in range(steps):
    act = str(rng.choice(["call", "call", "kernels", "gauss", "dt", "grad", "bilateral"]))
    B, C = int(rng.integers(1, 4)), int(rng.choice([1, 3]))
    H, W = int(rng.integers(30, 220)), int(rng.integers(30, 300))
    if MID and rng.random() < 0.5: H, W = int(rng.integers(220, 700)), int(rng.integers(300, 1000))
    x, _ = synthetic_blurry_batch(B, C, H, W, seed0=int(rng.integers(0, 10 ** 6)))
    err, tol, what = 0.0, 1.0, act
    if act == "call":
        _, kw, coef = _random_case(int(rng.integers(0, 10 ** 6)))
        if rng.random() < 0.35:                          # other kernel sizes (even ones off-centre, above 25 the large-kernel pass), the adaptive support
            kw["ker_size"] = int(rng.choice([7, 11, 12, 17, 24, 31, 40]))
            if min(H, W) < 2 * kw["ker_size"] + 2: kw.pop("ker_size")
        got, gi = polyblur_deblurring(torch.from_numpy(x).cuda(), return_info=True, **kw, **coef)
        want, wi = ref.polyblur_deblurring(x, return_info=True, **kw, **coef)
        same = all(np.array_equal(p["theta"], q["theta"]) for p, q in zip(gi, wi))
        err = float(np.abs(got.cpu().numpy() - want).max()) if same else 0.0       # (a flipped near-tie is the other sweeps' business)
        tol = 1e-4; what = "call %s" % kw
    elif act in ("kernels", "gauss"):
        method = str(rng.choice(["fft", "direct"])); bd = capi.PB_WRAP if method == "fft" else capi.PB_ZERO
        if act == "kernels":
            ks = np.stack([kernel(rng) for _ in range(B)]); buf = eng.set_kernels(ks)
        else:
            sg, rh, th = rng.uniform(0.3, 4, B).astype(np.float32), rng.uniform(0.3, 4, B).astype(np.float32), rng.uniform(0, np.pi, B).astype(np.float32)
            buf = eng.make_kernels(sg, rh, th); ks = eng.read_info(buf, B)["kernel"]
        sub = str(rng.choice(["inverse", "inverse_taper", "convolve2d", "edgetaper"]))
        xp = ref.replicate_pad(x, K // 2)
        if sub.startswith("inverse"):
            tp = sub == "inverse_taper"
            err = float(np.abs(eng.inverse_filter(x, buf, 6.0, 1.0, bd, edgetaping=tp) - ref.inverse_filtering_rank3(x, ks[:, None], 6.0, 1.0, do_edgetaper=tp, method=method)).max()); tol = 4e-5
        elif sub == "convolve2d":
            err = float(np.abs(eng.convolve2d(xp, buf, bd) - ref.convolve2d(xp, ks[:, None], method=method)).max()); tol = 2e-6
        else:
            err = float(np.abs(eng.edgetaper(xp, buf, bd) - ref.edgetaper(xp, ks[:, None], method=method)).max()); tol = 4e-6
        what = "%s %s %s" % (act, sub, method)
    elif act == "dt":
        ss, sr, N = float(rng.uniform(1, 40)), float(rng.uniform(0.2, 1)), int(rng.integers(1, 4))
        err = float(np.abs(eng.dt_recursive_filter(x, ss, sr, N) - ref.recursive_filter(x, ss, sr, N)).max()); tol = 5e-6
    elif act == "grad":
        gx, gy = eng.fourier_gradients(x); rx, ry = ref.spectral_gradients(x)
        err = max(float(np.abs(gx - rx).max()), float(np.abs(gy - ry).max())); tol = 2e-5
    else:
        err = float(np.abs(eng.bilateral5(x) - ref.bilateral_filter(x)).max()); tol = 3e-6
    count[act] = count.get(act, 0) + 1
    if not err < tol:
        bad += 1; print("step", t, what, (B, C, H, W), "err %.3e (tol %.1e)" % (err, tol), flush=True)
print("sequence of %d calls on one engine %s: %d outside tolerance" % (steps, count, bad))
