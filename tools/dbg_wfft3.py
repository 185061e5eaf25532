"""Debug: wave-private tile-spectrum body against the workgroup body on the same planes (convolve2d), bad-sample map."""
import sys, os, numpy as np, torch, ctypes as C
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import _capi as capi
from polyblur_amd.engine import Engine
os.environ["PB_FFT_BODY"] = "wg"; ewg = Engine(0)
os.environ["PB_FFT_BODY"] = "wave"; ewv = Engine(0)
sizes = [(1, 2184, 3864), (1, 1104, 1944), (1, 524, 724), (2, 1104, 1944)] if len(sys.argv) < 4 else [tuple(int(v) for v in sys.argv[1:4])]
for (B, H, W) in sizes:
    torch.manual_seed(1)
    xp = torch.rand(B, 3, H, W, device='cuda')
    outs = []
    for eng in (ewg, ewv, ewv):
        buf = eng.make_kernels([2.0] * B, [1.3] * B, [np.deg2rad(np.float32(66.0))] * B, support=0, name="dbg")
        op = torch.full_like(xp, 7.0)
        eng._check(eng.lib.pb_convolve2d(eng.ctx, C.c_void_p(xp.data_ptr()), C.c_void_p(op.data_ptr()), B, 3, H, W, buf.ptr, capi.PB_WRAP))
        torch.cuda.synchronize()
        outs.append(op.cpu().numpy())
    d = np.abs(outs[0] - outs[1]); d2 = np.abs(outs[1] - outs[2])
    bad = d > 1e-4
    print("size", B, H, W, "wg vs wave max %.3g nbad %d; wave run-to-run max %.3g nbad %d" % (d.max(), bad.sum(), d2.max(), (d2 > 0).sum()))
    if bad.any():
        idx = np.argwhere(bad)
        for pl in np.unique(idx[:, 0] * 3 + idx[:, 1]):
            m = bad[pl // 3, pl % 3]
            ys, xs = np.nonzero(m)
            print("  plane", pl, "nbad", m.sum(), "rows %d..%d cols %d..%d" % (ys.min(), ys.max(), xs.min(), xs.max()))
            # windows of 40x80 outputs (pairs): which pairs are affected
            py, px = ys // 40, xs // 80
            pairs = np.unique(py * 1000 + px)
            print("   pairs affected", len(pairs), "first", [(int(p) // 1000, int(p) % 1000) for p in pairs[:12]])
            p0 = pairs[0]; sel = (py * 1000 + px) == p0
            print("   in first pair: rows", np.unique(ys[sel] % 40).tolist(), "cols", np.unique(xs[sel] % 80).tolist()[:80])
            y, x = ys[sel][0], xs[sel][0]
            print("   sample", y, x, "wg %.6f wave %.6f second-run %.6f" % (outs[0][pl // 3, pl % 3, y, x], outs[1][pl // 3, pl % 3, y, x], outs[2][pl // 3, pl % 3, y, x]))
