#!/usr/bin/env python3
"""Stage stamps of the one-plan line transforms (lab build: tools/build_variant.sh lftrace "-DPB_EXPERIMENTAL -DPB_LINES_TRACE"
lines_fixed.hip; POLYBLUR_HIP_LIB=tools/_abl/lib_lftrace.so python tools/lines_fixed_trace.py).  Wall clock (100 MHz), thread 0 of
four workgroups of each launch: us after the earliest entry stamp of the launch."""
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
names = {0: ["entry", "first stage done", "barrier passed", "middle down", "centre", "middle up", "last stage: stores issued"],
         1: ["entry", "tile requests issued", "tile arrived", "first stage", "middle down", "centre", "middle up", "last stage + fold"]}
acc = []
for rep in range(6):
    polyblur_deblurring(d, n_iter=1, **KW)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    f = eng.lib.pb_debug_lines_fixed_trace; f.argtypes = [C.c_void_p]; f.restype = C.c_int
    assert f(buf) == 0
    if rep:
        acc.append(np.array(buf[:], dtype=np.float64).reshape(2, 4, 8))
t = np.mean(acc, axis=0)
for k, kname in enumerate(("rows (gray_rows_fixed_kernel)", "columns (cols_fixed_kernel)")):
    t0 = t[k, :, 0].min()
    print(kname)
    for w, wname in enumerate(("workgroup 0", "grid / 4", "grid / 2", "last")):
        print("   %-12s " % wname + "  ".join("%s %.1f" % (n, (t[k, w, i] - t0) / 100.0) for i, n in enumerate(names[k])))
# both launches on one clock: how long after the row kernel's last stamped store the column workgroups enter
r_last = t[0, :, 6].max(); c_entry = t[1, :, 0].min(); r_entry = t[0, :, 0].min()
print("rows: first entry -> last stamped 'stores issued' %.1f us; that stamp -> first column workgroup's entry %.1f us; columns: entry -> last stamp %.1f us"
      % ((r_last - r_entry) / 100.0, (c_entry - r_last) / 100.0, (t[1, :, 7].max() - c_entry) / 100.0))
# over ALL workgroups: earliest entry / latest exit (stores completed) of either launch, several repetitions
f2 = eng.lib.pb_debug_lines_fixed_span; f2.argtypes = [C.c_void_p, C.c_int]; f2.restype = C.c_int
spans = []
for rep in range(8):
    assert f2(None, 1) == 0
    polyblur_deblurring(d, n_iter=1, **KW)
    torch.cuda.synchronize()
    b4 = (C.c_ulonglong * 4)()
    assert f2(b4, 0) == 0
    if rep >= 2:
        v = np.array(b4[:], dtype=np.float64) / 100.0
        spans.append([v[1] - v[0], v[2] - v[1], v[3] - v[2]])
m = np.mean(spans, axis=0)
print("all workgroups: rows first entry -> last exit %.1f us; -> first column entry %.1f us; columns first entry -> last exit %.1f us" % tuple(m))
# per workgroup of the last repetition: entry and exit, by XCC
f3 = eng.lib.pb_debug_lines_fixed_wg; f3.argtypes = [C.c_void_p]; f3.restype = C.c_int
bw = (C.c_ulonglong * (2 * 2048 * 3))()
assert f3(bw) == 0
w = np.array(bw[:], dtype=np.float64).reshape(2, 2048, 3)
for k, (kname, n) in enumerate((("rows", 1080), ("columns", 240))):
    e, x, xcc = w[k, :n, 0] / 100.0, w[k, :n, 1] / 100.0, w[k, :n, 2].astype(int) & 15
    t0 = e.min()
    print("%s: entry  min %.1f  p50 %.1f  p90 %.1f  max %.1f | exit  min %.1f  p50 %.1f  p90 %.1f  max %.1f | duration  min %.1f  p50 %.1f  p90 %.1f  max %.1f"
          % (kname, 0.0, np.median(e) - t0, np.percentile(e, 90) - t0, e.max() - t0, x.min() - t0, np.median(x) - t0, np.percentile(x, 90) - t0, x.max() - t0,
             (x - e).min(), np.median(x - e), np.percentile(x - e, 90), (x - e).max()))
    for c in range(8):
        sel = xcc == c
        if sel.any():
            print("   xcc %d: %3d workgroups, entry max %.1f, exit p50 %.1f max %.1f, duration p50 %.1f max %.1f" % (c, sel.sum(), e[sel].max() - t0, np.median(x[sel]) - t0, x[sel].max() - t0, np.median((x - e)[sel]), (x - e)[sel].max()))
    late = np.argsort(x)[-8:]
    print("   latest exits: " + ", ".join("wg %d (xcc %d) entry %.1f exit %.1f" % (i, xcc[i], e[i] - t0, x[i] - t0) for i in late))
