"""Job timeline of the persistent tile-spectrum kernel (debug build -DPB_WF_TRACE): python tools/wf_jobs.py [poly|one]"""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import _capi as capi
from polyblur_amd.engine import get_engine
mode = sys.argv[1] if len(sys.argv) > 1 else "poly"
H, W, B = 2160, 3840, 1
eng = get_engine(0)
buf = eng.make_kernels([2.0] * B, [1.0] * B, [np.deg2rad(np.float32(30.0))] * B, support=0, name="bi")
x = torch.rand(B, 3, H, W, device='cuda'); o = torch.empty_like(x)
xp = torch.rand(B, 3, H + 24, W + 24, device='cuda'); op = torch.empty_like(xp)
f = eng.lib.pb_debug_wf_jobs; f.argtypes = [C.c_void_p, C.c_int]; f.restype = C.c_int
def run():
    if mode == "poly":
        eng._check(eng.lib.pb_inverse_filter(eng.ctx, C.c_void_p(x.data_ptr()), C.c_void_p(o.data_ptr()), capi.PB_F32, B, 3, H, W, buf.ptr, 6.0, 1.0, capi.PB_WRAP, 0, 0, None, None))
    else:
        eng._check(eng.lib.pb_convolve2d(eng.ctx, C.c_void_p(xp.data_ptr()), C.c_void_p(op.data_ptr()), B, 3, H + 24, W + 24, buf.ptr, capi.PB_WRAP))
    torch.cuda.synchronize()
run(); run()
assert f(None, 1) == 0
run()
NW, NS = 2048, 40
host = np.zeros((NW, NS, 3), np.uint64)
assert f(host.ctypes.data, 0) == 0
t = host.astype(np.int64)
valid = t[:, :, 1] > 0
t0 = t[:, :, 0][valid].min(); t1 = t[:, :, 1][valid].max()
print("mode", mode, "span %.1f us; jobs traced %d; jobs per wave min %d mean %.1f max %d" % ((t1 - t0) / 100.0, valid.sum(), valid.sum(1).min(), valid.sum(1).mean(), valid.sum(1).max()))
dur = (t[:, :, 1] - t[:, :, 0])[valid] / 100.0
print("job duration us: mean %.2f median %.2f p90 %.2f max %.2f" % (dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max()))
step = (t[:, :, 2] >> 28)
for s in range(3):
    m = valid & (step == s)
    if m.any():
        print(" step %d: jobs %d, first taken %.1f us, last done %.1f us, mean duration %.2f" % (s, m.sum(), (t[:, :, 0][m].min() - t0) / 100.0, (t[:, :, 1][m].max() - t0) / 100.0, ((t[:, :, 1] - t[:, :, 0])[m]).mean() / 100.0))
# gaps between consecutive jobs of a wave
gaps = []
for w in range(NW):
    n = valid[w].sum()
    for k in range(1, n): gaps.append((t[w, k, 0] - t[w, k - 1, 1]) / 100.0)
gaps = np.array(gaps)
print("gap between a wave's jobs us: mean %.2f median %.2f p90 %.2f max %.2f" % (gaps.mean(), np.median(gaps), np.percentile(gaps, 90), gaps.max()))
first = np.array([t[w, 0, 0] for w in range(NW) if valid[w, 0]]) - t0
last = np.array([t[w, valid[w].sum() - 1, 1] for w in range(NW) if valid[w, 0]]) - t0
print("first job taken us: min %.1f median %.1f max %.1f; last job done us: min %.1f median %.1f max %.1f" % (first.min() / 100., np.median(first) / 100., first.max() / 100., last.min() / 100., np.median(last) / 100., last.max() / 100.))
edges = np.linspace(t0, t1, 21)
live = [int(((t[:, :, 0] < b_) & (t[:, :, 1] > a_) & valid).sum()) for a_, b_ in zip(edges[:-1], edges[1:])]
print("jobs live per 5% slice:", live)
# duration by position in the wave's sequence
for k in range(0, 14):
    m = valid[:, k]
    if m.any(): print("  job #%d of a wave: mean duration %.2f us, mean start %.1f us" % (k, ((t[:, k, 1] - t[:, k, 0])[m]).mean() / 100.0, (t[:, k, 0][m].mean() - t0) / 100.0))
if mode == "one":
    j = (t[:, :, 2] & ((1 << 28) - 1)); pair = j % 2695; ty = pair // 49; px = pair % 49
    border = (ty == 0) | (ty >= 53) | (px == 0) | (px >= 47)
    d = (t[:, :, 1] - t[:, :, 0]) / 100.0
    print("interior jobs: %d mean %.1f us p90 %.1f; border jobs: %d mean %.1f us p90 %.1f" % ((valid & ~border).sum(), d[valid & ~border].mean(), np.percentile(d[valid & ~border], 90), (valid & border).sum(), d[valid & border].mean(), np.percentile(d[valid & border], 90)))
    slow = valid & (d > 60)
    print("slow jobs (>60us): %d, of which border %d; their start times us: min %.0f median %.0f" % (slow.sum(), (slow & border).sum(), (t[:, :, 0][slow].min() - t0) / 100.0, (np.median(t[:, :, 0][slow]) - t0) / 100.0))
    # was the job taken from the wave's own queue?
    per = (8085 + 7) // 8
    own = (j // per) == (np.arange(NW)[:, None] % 8)
    print("stolen jobs: %d, mean duration %.1f; own jobs mean duration %.1f" % ((valid & ~own).sum(), d[valid & ~own].mean() if (valid & ~own).any() else 0, d[valid & own].mean()))
    for lo in range(0, 320, 20):
        m = valid & ((t[:, :, 0] - t0) / 100.0 >= lo) & ((t[:, :, 0] - t0) / 100.0 < lo + 20)
        if m.any(): print("  jobs taken in [%3d,%3d) us: %5d, mean duration %.1f, border share %.2f" % (lo, lo + 20, m.sum(), d[m].mean(), border[m].mean()))
