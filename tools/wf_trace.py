"""Phase timeline of the wave-private tile-spectrum body (debug build: PB_EXTRA_FLAGS=-DPB_WF_TRACE python -m polyblur_amd.build --force)."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import _capi as capi
from polyblur_amd.engine import get_engine
H, W, B = 2160, 3840, 1
eng = get_engine(0)
x = torch.rand(B, 3, H, W, device='cuda'); o = torch.empty_like(x)
buf = eng.make_kernels([2.0] * B, [1.0] * B, [np.deg2rad(np.float32(30.0))] * B, support=0, name="bi")
ms = eng.time_inner_loop(x.data_ptr(), o.data_ptr(), capi.PB_F32, x.shape, buf.ptr, 6, 1, capi.PB_WRAP, 3)
torch.cuda.synchronize()
NW, NS = 8192, 12
host = np.zeros((NW, NS), np.uint64)
f = eng.lib.pb_debug_wf_trace; f.argtypes = [C.c_void_p, C.c_int]; f.restype = C.c_int
assert f(host.ctypes.data, NW) == 0
t = host.astype(np.int64)
ok = (t[:, 10] > 0) & (t[:, 1] > 0)
t = t[ok]
d = np.diff(t[:, :11], axis=1)
names = ["entry->job", "job->loads issued", "col fwd (incl load wait)", "transpose 1", "kh + row stage 1", "centre + inv stage 1", "transpose 2",
         "x req + inv stage 2", "finish groups", "drain stores"]
print("ms/poly %.4f; %d waves traced; mean cycles per phase (total %.0f):" % (ms, len(t), (t[:, 10] - t[:, 0]).mean()))
for n, m, md in zip(names, d.mean(0), np.median(d, 0)):
    print("  %-28s mean %8.0f  median %8.0f" % (n, m, md))
print("start spread: first wave starts", (t[:, 0] - t[:, 0].min())[:16])
