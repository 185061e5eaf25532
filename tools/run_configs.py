"""Time the BASELINE.json configurations on ONE GPU (per-GPU share of the multi-GPU configs).
    python tools/run_configs.py [--small]"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, '.')
from polyblur_amd import polyblur_deblurring
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch

KW = dict(c=0.362, b=0.468, alpha=6, beta=1)
small = "--small" in sys.argv


def batch(b, h, w, dtype):
    nd = min(b, 2)
    x, _ = synthetic_blurry_batch(nd, 3, h, w, seed0=4242)
    x = np.concatenate([x] * ((b + nd - 1) // nd))[:b]
    return torch.from_numpy(x).cuda().to(dtype).contiguous()


def run(name, x, reps=5, **kw):
    eng = get_engine(0)
    for _ in range(2):
        out = polyblur_deblurring(x, **KW, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = polyblur_deblurring(x, **KW, **kw)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    B, _, H, W = x.shape
    rec = dict(config=name, shape=list(x.shape), dtype=str(x.dtype).split('.')[-1], ms=round(ms, 3),
               mp_per_s=round(B * H * W / 1e6 / (ms * 1e-3), 1), workspace_MB=round(eng.workspace_bytes() / 1e6, 1),
               finite=bool(torch.isfinite(out.float()).all()), options=kw)
    print(json.dumps(rec), flush=True)


run("cfg2: 1x4K fp32 n_iter=3", batch(1, 2160, 3840, torch.float32), n_iter=3)
b3 = 8 if small else 64
run("cfg3: %dx1080p fp16 n_iter=3 + halo + domain-transform prefilter" % b3, batch(b3, 1080, 1920, torch.float16), reps=3,
    n_iter=3, remove_halo=True, prefiltering=True, prefilter="domain_transform")
run("cfg3': same with the reference's live prefilter (bilateral)", batch(b3, 1080, 1920, torch.float16), reps=3,
    n_iter=3, remove_halo=True, prefiltering=True)
b4 = 8 if small else 32
run("cfg4: %dx1080p fp32 n_iter=3 (per-GPU share of 256 over 8 GPUs)" % b4, batch(b4, 1080, 1920, torch.float32), reps=3, n_iter=3)
run("cfg5: 1x8K fp16 n_iter=5 (per-GPU share of 8 over 8 GPUs)", batch(1, 4320, 7680, torch.float16), reps=3, n_iter=5)
run("cfg1: peacock-size 1x700x500 fp32 n_iter=3", batch(1, 500, 700, torch.float32), reps=20, n_iter=3)
