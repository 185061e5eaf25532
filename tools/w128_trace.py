"""Where wave 0 of a conv_w128 workgroup spends its cycles (lab build: tools/build_variant.sh w128trace
"-DPB_EXPERIMENTAL -DPB_W128_TRACE" conv_w128.hip; POLYBLUR_HIP_LIB=tools/_abl/lib_w128trace.so python tools/w128_trace.py)."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from polyblur_amd import polyblur_deblurring
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch
KW = dict(n_iter=1, c=0.362, b=0.468, alpha=6, beta=1)
x = torch.from_numpy(synthetic_blurry_batch(1, 3, 2160, 3840, seed0=int(os.environ.get("SEED", "0")))[0]).cuda()
eng = get_engine(0)
f = eng.lib.pb_debug_w128_trace; f.argtypes = [C.c_void_p]; f.restype = C.c_int
host = np.zeros((8192, 12), np.uint64)
for _ in range(3): polyblur_deblurring(x, **KW)
torch.cuda.synchronize(); f(host.ctypes.data)
polyblur_deblurring(x, **KW); torch.cuda.synchronize()
assert f(host.ctypes.data) == 0
t = host.astype(np.float64); t = t[t[:, 8] > 0]
names = ["entry -> job found", "window loaded (DMA + LDS reads + radix-2)", "column transform", "transpose + rows' radix-2", "row transform x spectrum x inverse",
         "transpose back", "inverse column transform", "epilogue (stores issued)"]
print("workgroups with a job: %d" % len(t))
tot = (t[:, 8] - t[:, 0]).mean()
for i, n in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    print("  %-48s %7.0f cycles  (%4.1f %%)   p10 %7.0f  p90 %7.0f" % (n, d.mean(), 100 * d.mean() / tot, np.percentile(d, 10), np.percentile(d, 90)))
print("  total %.0f cycles" % tot)
rt0 = t[:, 9].min(); dur = (t[:, 10] - t[:, 9]) / 100.0
print("realtime: launch span %.1f us; workgroup duration mean %.1f us (p10 %.1f, p90 %.1f); last workgroup starts at %.1f us; mean concurrency %.0f workgroups" % (
    (t[:, 10].max() - rt0) / 100.0, dur.mean(), np.percentile(dur, 10), np.percentile(dur, 90), (t[:, 9].max() - rt0) / 100.0, dur.sum() / ((t[:, 10].max() - rt0) / 100.0)))
st = np.sort((t[:, 9] - rt0) / 100.0)
print("workgroup starts (us): " + " ".join("%.0f" % st[int(q * (len(st) - 1))] for q in (0, .1, .2, .3, .4, .5, .6, .7, .8, .9, 1)))
print("shader clock %.2f GHz" % ((t[:, 8] - t[:, 0]) / (dur * 1e3)).mean())
