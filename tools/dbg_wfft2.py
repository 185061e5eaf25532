import sys, os, numpy as np
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from oracle import polyblur_ref as ref
from polyblur_amd import _capi as capi
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch
eng = get_engine(0)
eng.set_dense_eval("auto", 0)
B, H, W = 1, 152, 216
sg, rh, deg = 1.3, 0.8, 60.0
x, _ = synthetic_blurry_batch(B, 3, H, W, seed0=11)
xp = np.pad(x, ((0, 0), (0, 0), (12, 12), (12, 12)), mode="edge")
th = np.float32(deg) * np.float32(np.pi) / np.float32(180)
k = ref.gaussian_kernel_2d([th] * B, [sg] * B, [rh] * B)
buf = eng.make_kernels([sg] * B, [rh] * B, [th] * B, support=capi.PB_SUPPORT_ADAPTIVE)
eng.buffer("np.out", xp.nbytes).upload(np.full(xp.shape, 7.0, np.float32))
out = eng.convolve2d(xp, buf, capi.PB_ZERO)
want = ref.convolve2d(xp, k[:, None], method="direct")
d = np.abs(out - want)[0, 0]
print("convolve2d err", d.max())
bad = d > 1e-4
rows = np.unique(np.argwhere(bad)[:, 0]); cols = np.unique(np.argwhere(bad)[:, 1])
print("bad rows", rows.tolist()); print("bad cols", cols.tolist())
o = out[0, 0]; w = want[0, 0]
ys, xs = np.argwhere(bad)[:6].T
for y, x in zip(ys, xs):
    cands = {}
    for dy in (-48, -32, -16, 16, 32):
        for dx in (-48, 0, 48):
            yy, xx = y + dy, x + dx
            if 0 <= yy < w.shape[0] and 0 <= xx < w.shape[1] and abs(w[yy, xx] - o[y, x]) < 1e-5:
                cands[(dy, dx)] = float(w[yy, xx])
    # search whole want plane for the value
    hits = np.argwhere(np.abs(want[0] - o[y, x]) < 2e-6)[:5].tolist()
    print("bad", y, x, "got %.9e want %.6f" % (o[y, x], w[y, x]), "matches", cands, "hits(c,y,x)", hits, "xp %.6f" % xp[0, 0, y, x])
import collections
byrow = collections.defaultdict(list)
for y, x in np.argwhere(bad): byrow[int(y)].append(int(x))
for y in sorted(byrow): print(y, byrow[y])
