#!/usr/bin/env python3
"""Phase stamps of the estimation's parameter kernel (debug build: tools/build_variant.sh ptrace "-DPB_EXPERIMENTAL -DPB_PARAMS_TRACE" estimate.hip
conv_fft.hip; POLYBLUR_HIP_LIB=tools/_abl/lib_ptrace.so python tools/params_trace.py).  Shader-clock cycles, first workgroup."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from polyblur_amd import polyblur_deblurring
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch

eng = get_engine(0)
img, _ = synthetic_blurry_batch(1, 3, 2160, 3840, seed0=5)
d = torch.from_numpy(img).cuda()
KW = dict(c=0.362, b=0.468, alpha=6, beta=1)
names = {0: "entry", 1: "maxima + range folded", 2: "interpolated", 3: "argmin, sigma, rho", 4: "taps + sum", 5: "marginals", 6: "acorr, gtaps, residual terms",
         7: "residual sum", 8: "phases", 20: "khat: taps copied, symmetry", 24: "khat: marginals", 25: "khat: convolution powers", 26: "khat: tail radii", 23: "khat: halos measured, form chosen", 27: "khat128: tables in LDS", 28: "khat128: first sums", 22: "khat: spectrum stored"}
acc = {}
for rep in range(6):
    polyblur_deblurring(d, n_iter=1, **KW)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    f = eng.lib.pb_debug_params_trace; f.argtypes = [C.c_void_p]; f.restype = C.c_int
    assert f(buf) == 0
    t = np.array(buf[:], dtype=np.int64)
    if rep:
        for k in names: acc.setdefault(k, []).append(int(t[k] - t[0]))
for k in sorted(names, key=lambda k: np.mean(acc[k])):
    if np.mean(acc[k]) >= 0:
        print("%-34s %8.0f cycles after entry" % (names[k], np.mean(acc[k])))
