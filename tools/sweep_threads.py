"""Two host threads, an engine (context) each, on one device, calling at the same time: every result against the oracle
(computed beforehand).  python tools/sweep_threads.py [calls per thread]"""
import sys, threading, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import polyblur_ref as ref
from polyblur_amd.engine import Engine
from polyblur_amd.synthetic import synthetic_blurry_batch
from test_gpu_parity import _random_case
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
work = {}
for tid in range(2):
    rng = np.random.default_rng(100 + tid); items = []
    for i in range(n):
        _, kw, coef = _random_case(int(rng.integers(0, 10 ** 6)))
        B, C, H, W = int(rng.integers(1, 4)), int(rng.choice([1, 3])), int(rng.integers(30, 260)), int(rng.integers(30, 330))
        x, _ = synthetic_blurry_batch(B, C, H, W, seed0=int(rng.integers(0, 10 ** 6)))
        want, wi = ref.polyblur_deblurring(x, return_info=True, **kw, **coef)
        items.append((x, kw, coef, want, wi))
    work[tid] = items
bad = [0, 0]
def run(tid):
    eng = Engine(0)
    for x, kw, coef, want, wi in work[tid]:
        k2 = dict(kw); pf = k2.pop("prefilter", "bilateral")
        from polyblur_amd import _capi as capi
        opts = eng.make_options(**{**{k: v for k, v in k2.items() if k not in ("method", "prefiltering")}, **coef},
                                boundary=capi.PB_WRAP if kw["method"] == "fft" else capi.PB_ZERO,
                                prefilter=(capi.PB_PREFILTER_NONE if not kw.get("prefiltering") else (capi.PB_PREFILTER_BILATERAL if pf == "bilateral" else capi.PB_PREFILTER_DOMAIN_TRANSFORM)))
        out, infos = eng.polyblur(x, opts, want_info=True)
        same = all(np.array_equal(np.asarray(infos[k]["theta"], np.float32).reshape(-1), np.asarray(wi[k]["theta"], np.float32).reshape(-1)) for k in range(len(wi)))
        if same and not np.abs(out - want).max() < 1e-4: bad[tid] += 1
ts = [threading.Thread(target=run, args=(t,)) for t in range(2)]
for t in ts: t.start()
for t in ts: t.join()
print("two threads x %d calls at once: %s outside tolerance" % (n, bad))
