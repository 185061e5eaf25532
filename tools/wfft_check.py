#!/usr/bin/env python3
"""Parity and timing of the tile-spectrum wave body with per-axis run-time halos and the general one-pass polynomial
(csrc/conv_wfft.hip, csrc/khat.h).  GPU box only.

    python tools/wfft_check.py [parity] [timing] [call]

parity: eng.inverse_filter against the oracle for a spread of kernels, shapes and dtypes, through the default context
        (PB_POLY1=2) and through the three-step form (PB_POLY1=0); prints the body each image took.
timing: one 4K polynomial (pb_time_inner_loop) per kernel, both contexts.
call:   the whole 4K headline call, both contexts."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import polyblur_ref as ref
from polyblur_amd import _capi as capi
from polyblur_amd.engine import Engine
from polyblur_amd.synthetic import synthetic_blurry_batch

what = set(sys.argv[1:]) or {"parity", "timing", "call"}


def make(mode):
    old = os.environ.get("PB_POLY1")
    os.environ["PB_POLY1"] = str(mode)
    try:
        return Engine(0)
    finally:
        if old is None:
            del os.environ["PB_POLY1"]
        else:
            os.environ["PB_POLY1"] = old


one, three = make(int(os.environ.get("WFFT_CHECK_MODE", "3"))), make(0)
KERNELS = [  # theta deg, sigma, rho
    (66.0, 2.095, 1.314), (66.0, 1.656, 1.009), (66.0, 1.24, 0.625), (0.0, 2.0, 1.0), (0.0, 1.4, 0.9), (30.0, 0.65, 0.40),
    (0.0, 0.3, 0.3), (45.0, 3.0, 1.0), (0.0, 4.0, 4.0), (90.0, 1.2, 0.5), (120.0, 0.9, 0.5),
]
worst = 0.0
if "parity" in what:
    for shape, dtype in (((1, 3, 1080, 1920), np.float32), ((2, 1, 301, 517), np.float32), ((1, 3, 1080, 1920), np.float16),
                         ((1, 3, 150, 210), np.float32)):
        B = shape[0]
        x, _ = synthetic_blurry_batch(*shape, seed0=91)
        xin = x.astype(dtype)
        for deg, sg, rh in KERNELS:
            th = [np.float32(np.deg2rad(deg))] * B
            res = []
            for eng in (one, three):
                buf = eng.make_kernels([sg] * B, [rh] * B, th, support=capi.PB_SUPPORT_FULL)
                info = eng.read_info(buf, B)
                out = eng.inverse_filter(xin, buf, 6.0, 1.0, capi.PB_WRAP).astype(np.float32)
                res.append((out, eng.body_selection(B)[0].tolist()))
            want = ref.inverse_filtering_rank3(xin.astype(np.float32), info["kernel"][:, None], 6.0, 1.0, method="fft")
            e1, e3 = float(np.abs(res[0][0] - want).max()), float(np.abs(res[1][0] - want).max())
            worst = max(worst, e1 if dtype == np.float32 else 0.0)
            print("%-18s %-7s theta %5.1f sigma %.3f rho %.3f | one-pass ctx sel %s err %.2e | three-step ctx sel %s err %.2e"
                  % (shape, np.dtype(dtype).name, deg, sg, rh, res[0][1], e1, res[1][1], e3), flush=True)
    print("worst fp32 error of the default context: %.2e" % worst)

if "timing" in what:
    x4 = torch.rand(1, 3, 2160, 3840, device="cuda")
    o4 = torch.empty_like(x4)
    for deg, sg, rh in KERNELS:
        row = []
        for eng in (one, three):
            eng.set_stream(torch.cuda.current_stream(0).cuda_stream)
            buf = eng.make_kernels([sg], [rh], [np.float32(np.deg2rad(deg))], support=capi.PB_SUPPORT_FULL, name="t.info")
            ms = min(eng.time_inner_loop(x4.data_ptr(), o4.data_ptr(), capi.PB_F32, x4.shape, buf.ptr, 6, 1, capi.PB_WRAP, 20) for _ in range(3))
            row.append((ms, eng.body_selection(1)[0].tolist()))
        print("4K polynomial theta %5.1f sigma %.3f rho %.3f | default %.4f ms sel %s | three-step %.4f ms sel %s | %.2fx, %4.0f GB/s algorithmic"
              % (deg, sg, rh, row[0][0], row[0][1], row[1][0], row[1][1], row[1][0] / row[0][0],
                 8.0 * 4 * x4.numel() / (row[0][0] * 1e-3) / 1e9), flush=True)

if "call" in what:
    img, _ = synthetic_blurry_batch(1, 3, 2160, 3840)
    d = torch.from_numpy(img).cuda()
    kw = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1)
    o = one.make_options(**kw)
    outs = []
    for eng, name in ((one, "default"), (three, "three-step")):
        eng.set_stream(torch.cuda.current_stream(0).cuda_stream)
        out = torch.empty_like(d)
        for _ in range(3):
            eng.polyblur_ptr(d.data_ptr(), out.data_ptr(), capi.PB_F32, d.shape, o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.polyblur_ptr(d.data_ptr(), out.data_ptr(), capi.PB_F32, d.shape, o)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        outs.append(out.cpu().numpy())
        print("whole 4K call, %-10s %.4f ms = %.0f MP/s, last selection %s" % (name, ms, 3840 * 2160 / 1e6 / (ms * 1e-3), eng.body_selection(1)[0].tolist()), flush=True)
    print("one-pass vs three-step max abs %.2e" % float(np.abs(outs[0] - outs[1]).max()))
    want = ref.polyblur_deblurring(img, **kw)
    print("default vs oracle %.2e, three-step vs oracle %.2e" % (float(np.abs(outs[0] - want).max()), float(np.abs(outs[1] - want).max())))
