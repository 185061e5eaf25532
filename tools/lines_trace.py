#!/usr/bin/env python3
"""Stage stamps of the two line transforms of the estimation (lab build: tools/build_variant.sh ltrace
"-DPB_EXPERIMENTAL -DPB_LINES_TRACE" estimate.hip; POLYBLUR_HIP_LIB=tools/_abl/lib_ltrace.so python tools/lines_trace.py).
Wall clock (100 MHz), thread 0 of four workgroups of each launch: us after the earliest entry stamp of the launch."""
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
names = ["entry", "first stage issued", "first stage: barrier passed", "middle stage(s) down", "centre stage", "middle stage(s) up", "last stage + epilogue"]
acc = []
for rep in range(6):
    polyblur_deblurring(d, n_iter=1, **KW)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    f = eng.lib.pb_debug_lines_trace; f.argtypes = [C.c_void_p]; f.restype = C.c_int
    assert f(buf) == 0
    if rep:
        acc.append(np.array(buf[:], dtype=np.float64).reshape(2, 4, 8))
t = np.mean(acc, axis=0)
for k, kname in enumerate(("rows (gray_rows_kernel)", "columns (grad_cols_kernel)")):
    t0 = t[k, :, 0].min()
    print(kname)
    for w, wname in enumerate(("workgroup 0", "grid / 4", "grid / 2", "last")):
        print("   %-12s " % wname + "  ".join("%s %.1f" % (names[i], (t[k, w, i] - t0) / 100.0) for i in range(7)) +
              ("  [first stage's requests served %.1f]" % ((t[k, w, 7] - t0) / 100.0) if t[k, w, 7] > 0 else ""))
