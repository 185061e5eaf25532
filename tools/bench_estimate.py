"""Per-stage time of the blur estimation (gray, row / column spectral derivative, parameters) on resident images.

    python tools/bench_estimate.py [--shape B,C,H,W] [--reps 30]

Times come from the library's own per-launch events (pb_profile_*), one estimation per image per repetition.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from polyblur_amd import _capi as capi
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="1,3,2160,3840")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--dtype", default="f32")
    args = ap.parse_args()
    B, C, H, W = (int(v) for v in args.shape.split(","))
    x, _ = synthetic_blurry_batch(min(B, 4), C, H, W, seed0=20260929)
    x = np.concatenate([x] * ((B + len(x) - 1) // len(x)))[:B]
    xt = torch.from_numpy(x).cuda()
    if args.dtype == "f16":
        xt = xt.half()
    out = torch.empty_like(xt)
    eng = get_engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    opts = eng.make_options(n_iter=1, c=0.362, b=0.468, alpha=6, beta=1)
    dt = capi.PB_F16 if args.dtype == "f16" else capi.PB_F32
    for _ in range(3):
        eng.polyblur_ptr(xt.data_ptr(), out.data_ptr(), dt, xt.shape, opts)
    eng.synchronize()
    eng.profile_begin()
    for _ in range(args.reps):
        eng.polyblur_ptr(xt.data_ptr(), out.data_ptr(), dt, xt.shape, opts)
    eng.synchronize()
    prof = eng.profile_end()
    tot = 0.0
    for tag in ("gray", "grad_rows", "grad_cols", "params"):
        ms, n = prof[tag]
        us = 1e3 * ms / args.reps
        tot += us
        print("%-10s %8.1f us per estimation (%d launches)" % (tag, us, n // args.reps))
    print("%-10s %8.1f us   (%s %s)" % ("sum", tot, args.shape, args.dtype))


if __name__ == "__main__":
    main()
