"""Quick inner-loop timing for kernel tuning: python tools/bench_inner.py [H W B]"""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import _capi as capi
from polyblur_amd.engine import get_engine
only = None
if "--only" in sys.argv:
    k = sys.argv.index("--only"); only = sys.argv[k + 1]; del sys.argv[k:k + 2]
H, W, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2160, 3840, 1)
eng = get_engine(0)
x = torch.rand(B, 3, H, W, device='cuda')
o = torch.empty_like(x)
s = 4
for name, th, sg, rh, sup in (("rank1 full", 0., 2., 1., 0), ("general full", 30., 2., 1., 0), ("general adaptive s1", 30., 1., .6, 1),
                              ("general adaptive s.6", 30., .6, .4, 1)):
    if only and only not in name:
        continue
    buf = eng.make_kernels([sg] * B, [rh] * B, [np.deg2rad(np.float32(th))] * B, support=sup, name="bi")
    ms = min(eng.time_inner_loop(x.data_ptr(), o.data_ptr(), capi.PB_F32, x.shape, buf.ptr, 6, 1, capi.PB_WRAP, 20) for _ in range(3))
    gbs = 8.0 * s * x.numel() / (ms * 1e-3) / 1e9
    print("%-22s %.4f ms/poly  %7.0f GB/s alg  %.1f%% of 8TB/s  %8.0f MP/s" % (name, ms, gbs, gbs / 80, B * H * W / 1e6 / (ms * 1e-3)))
