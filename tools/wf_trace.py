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
# ONE launch to stamp: a convolve2d of the padded planes (the launches above left their stamps behind: overwritten here)
xp = torch.rand(B, 3, H + 24, W + 24, device='cuda'); op = torch.empty_like(xp)
for rep in range(2):
    eng.lib.pb_debug_wf_trace_clear(); torch.cuda.synchronize()
    eng._check(eng.lib.pb_convolve2d(eng.ctx, C.c_void_p(xp.data_ptr()), C.c_void_p(op.data_ptr()), B, 3, H + 24, W + 24, buf.ptr, capi.PB_WRAP))
    torch.cuda.synchronize()
NW, NS = 2048, 14
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
rt0 = t[:, 12].min(); rt1 = t[:, 13].max()
print("realtime (100 MHz ticks): kernel span %d ticks = %.1f us; sum of wave durations %.1f us -> mean concurrency %.0f of 2048 slots" % (
    rt1 - rt0, (rt1 - rt0) / 100.0, (t[:, 13] - t[:, 12]).sum() / 100.0, (t[:, 13] - t[:, 12]).sum() / max(rt1 - rt0, 1)))
edges = np.linspace(rt0, rt1, 21)
print("live waves per 5% slice:", [int(((t[:, 12] < b_) & (t[:, 13] > a_)).sum()) for a_, b_ in zip(edges[:-1], edges[1:])])
st = np.sort(t[:, 12] - rt0)
print("wave start times (us): #0 %.1f #512 %.1f #1024 %.1f #2047 %.1f" % tuple(st[i] / 100.0 for i in (0, 512, 1024, -1)))
print("wave duration us: mean %.2f median %.2f p90 %.2f max %.2f" % tuple(np.percentile((t[:, 13] - t[:, 12]) / 100.0, q) if q >= 0 else (t[:, 13] - t[:, 12]).mean() / 100.0 for q in (-1, 50, 90, 100)))
