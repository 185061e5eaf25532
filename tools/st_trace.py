"""Where a wave of the strip body spends its cycles (debug build -DPB_ST_TRACE): python tools/st_trace.py"""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import _capi as capi
from polyblur_amd.engine import get_engine
H, W, B = 2160, 3840, 1
eng = get_engine(0)
buf = eng.make_kernels([2.0] * B, [1.0] * B, [0.0] * B, support=0, name="bi")
x = torch.rand(B, 3, H, W, device='cuda'); o = torch.empty_like(x)
ms = eng.time_inner_loop(x.data_ptr(), o.data_ptr(), capi.PB_F32, x.shape, buf.ptr, 6, 1, capi.PB_WRAP, 5)
torch.cuda.synchronize()
host = np.zeros((4096, 6), np.uint64)
f = eng.lib.pb_debug_st_trace; f.argtypes = [C.c_void_p]; f.restype = C.c_int
assert f(host.ctypes.data) == 0
t = host.astype(np.float64); t = t[t[:, 3] > 0]
print("ms/poly %.4f; waves %d; steps per wave %.0f; cycles per step: total %.0f, vmcnt wait %.0f, LDS reads %.0f" % (
    ms, len(t), t[:, 3].mean(), (t[:, 2] / t[:, 3]).mean(), (t[:, 0] / t[:, 3]).mean(), (t[:, 1] / t[:, 3]).mean()))
rt0, rt1 = t[:, 4].min(), t[:, 5].max()
dur = (t[:, 5] - t[:, 4]) / 100.0
print("realtime: kernel span %.1f us; wave duration mean %.1f us max %.1f; wave starts: median %.1f us, last %.1f us; mean concurrency %.0f waves" % (
    (rt1 - rt0) / 100.0, dur.mean(), dur.max(), np.median(t[:, 4] - rt0) / 100.0, (t[:, 4].max() - rt0) / 100.0, dur.sum() / ((rt1 - rt0) / 100.0)))
print("shader cycles per wave / realtime us -> clock %.2f GHz" % ((t[:, 2] / (dur * 1e3)).mean()))
