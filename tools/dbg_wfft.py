"""Debug: the wave-private tile-spectrum body against the oracle on shapes that take its all-16-byte path."""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from oracle import polyblur_ref as ref
from polyblur_amd import _capi as capi
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch
eng = get_engine(0)
eng.set_dense_eval("auto", 0)
for (B, H, W) in [(1, 152, 216), (2, 152, 216), (1, 240, 328), (1, 150, 210)]:
    for (sg, rh, deg) in [(3.0, 2.0, 66.0), (1.3, 0.8, 60.0), (0.6, 0.4, 24.0)]:
        for sup in (capi.PB_SUPPORT_FULL, capi.PB_SUPPORT_ADAPTIVE):
            x, _ = synthetic_blurry_batch(B, 3, H, W, seed0=11)
            th = np.float32(deg) * np.float32(np.pi) / np.float32(180)
            k = ref.gaussian_kernel_2d([th] * B, [sg] * B, [rh] * B)
            buf = eng.make_kernels([sg] * B, [rh] * B, [th] * B, support=sup)
            info = eng.read_info(buf, B)
            for bnd, m in ((capi.PB_WRAP, "fft"), (capi.PB_ZERO, "direct")):
                out = eng.inverse_filter(x, buf, 6.0, 1.0, bnd)
                want = ref.inverse_filtering_rank3(x, k[:, None], 6.0, 1.0, method=m)
                d = np.abs(out - want)
                bad = np.argwhere(d > 1e-4)
                print(B, H, W, sg, rh, deg, "sup", sup, m, "R", info["radius"], "err %.3g" % d.max(), "nbad", len(bad), bad[:3].tolist() if len(bad) else "")
