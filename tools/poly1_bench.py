#!/usr/bin/env python3
"""One-pass polynomial experiment (verdict item 4; env PB_POLY1=1, read when the context is created): the reference's 'fft'
form of the deconvolution is ONE filter a3 K^3 + a2 K^2 + a1 K + b (deblurring.py:139-169); a kernel within the 4-sample
halo has a composite of halo 12, so its three Horner launches become one window pass with the polynomial's spectrum.

Run once per setting (the flag is per context):  PB_POLY1=0 python tools/poly1_bench.py ; PB_POLY1=1 python tools/poly1_bench.py
Prints, per case: parity of the non-blind step against the oracle (1080p), the time of the three-step inner loop at 4K
(pb_time_inner_loop), and the whole 4K call under the adaptive policy.  GPU box only."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import polyblur_ref as ref
from polyblur_amd import _capi as capi, polyblur_deblurring
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch

eng = get_engine(0)
mode = os.environ.get("PB_POLY1", "0")
KW = dict(c=0.362, b=0.468, alpha=6, beta=1)
CASES = [(30.0, 0.65, 0.40), (75.0, 0.55, 0.35), (30.0, 0.45, 0.30), (30.0, 1.0, 0.6), (30.0, 2.0, 1.0)]

x, _ = synthetic_blurry_batch(1, 3, 1080, 1920, seed0=77)
for deg, sg, rh in CASES:
    for support, sname in ((capi.PB_SUPPORT_ADAPTIVE, "adaptive"), (capi.PB_SUPPORT_FULL, "full")):
        th = np.array([np.deg2rad(deg)], np.float32)
        buf = eng.make_kernels(np.array([sg], np.float32), np.array([rh], np.float32), th, support=support, name="p1.info")
        rec = eng.read_info(buf, 1)
        got = eng.inverse_filter(x, buf, KW["alpha"], KW["beta"], boundary=capi.PB_WRAP)
        want = ref.inverse_filtering_rank3(x, rec["kernel"][:, None], KW["alpha"], KW["beta"], method="fft")
        err = float(np.abs(got - want).max())
        # 4K timing of the inner loop
        x4 = torch.rand(1, 3, 2160, 3840, device="cuda")
        o4 = torch.empty_like(x4)
        eng.set_stream(torch.cuda.current_stream(0).cuda_stream)
        ms = eng.time_inner_loop(x4.data_ptr(), o4.data_ptr(), capi.PB_F32, x4.shape, buf.ptr, KW["alpha"], KW["beta"], capi.PB_WRAP, 20)
        print("PB_POLY1=%s  theta %4.0f sigma %.2f rho %.2f %-8s radius %2d separable %d | 1080p max err vs oracle %.2e | 4K polynomial %.4f ms"
              % (mode, deg, sg, rh, sname, int(rec["radius"][0]), int(rec["separable"][0]), err, ms), flush=True)

# the whole call: a sharp-ish 4K image whose later iterations estimate small kernels
for seed in (5, 6):
    img, _ = synthetic_blurry_batch(1, 3, 2160, 3840, seed0=seed)
    d = torch.from_numpy(img).cuda()
    for support in ("adaptive", "full"):
        out, infos = polyblur_deblurring(d, n_iter=3, support=support, return_info=True, **KW)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = polyblur_deblurring(d, n_iter=3, support=support, **KW)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        print("PB_POLY1=%s  whole 4K call seed %d %-8s %.4f ms  radii %s  checksum %.6f" % (
            mode, seed, support, ms, [int(i["radius"][0]) for i in infos], float(out.double().mean())), flush=True)

# a mildly blurred 4K image (the method's own use case): estimates like sigma 0.6 / rho 0.3 at an oblique angle
rng = np.random.default_rng(83)
x = rng.random((1, 3, 2160, 3840), dtype=np.float32)
x = np.clip(ref.convolve2d(x, ref.gaussian_kernel_2d([np.float32(0.6)], [0.9], [0.5]), method="fft"), 0, 1).astype(np.float32)
d = torch.from_numpy(x).cuda()
for support, c in (("adaptive", 0.4), ("full", 0.4), ("adaptive", 0.2), ("full", 0.2)):   # c = 0.2: the clamped isotropic estimate, rank-1
    kw = dict(n_iter=3, c=c, b=0.468, alpha=6, beta=1, support=support)
    out, infos = polyblur_deblurring(d, return_info=True, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        out = polyblur_deblurring(d, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print("PB_POLY1=%s  mildly blurred 4K %-8s %.4f ms  sigma/rho %s radii %s" % (
        mode, support, ms, [(round(float(i["sigma"][0]), 2), round(float(i["rho"][0]), 2)) for i in infos], [int(i["radius"][0]) for i in infos]), flush=True)
