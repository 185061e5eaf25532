"""Timing of ONE reblurring pass (pb_convolve2d on padded 4K planes, dense kernel) and of the polynomial."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import _capi as capi
from polyblur_amd.engine import get_engine
H, W, B = 2160, 3840, 1
eng = get_engine(0)
buf = eng.make_kernels([2.0] * B, [1.0] * B, [np.deg2rad(np.float32(30.0))] * B, support=0, name="bi")
xp = torch.rand(B, 3, H + 24, W + 24, device='cuda'); op = torch.empty_like(xp)
def once():
    eng._check(eng.lib.pb_convolve2d(eng.ctx, C.c_void_p(xp.data_ptr()), C.c_void_p(op.data_ptr()), B, 3, H + 24, W + 24, buf.ptr, capi.PB_WRAP))
for _ in range(3): once()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    e0.record()
    for _ in range(20): once()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
x = torch.rand(B, 3, H, W, device='cuda'); o = torch.empty_like(x)
ms = min(eng.time_inner_loop(x.data_ptr(), o.data_ptr(), capi.PB_F32, x.shape, buf.ptr, 6, 1, capi.PB_WRAP, 20) for _ in range(3))
print("%s: one pass %.1f us; polynomial %.1f us (%.1f per step)" % (os.environ.get("POLYBLUR_HIP_LIB", "default"), best * 1e3, ms * 1e3, ms * 1e3 / 3))
