"""Launch-bound small images: eager launches vs hipGraph replay.  python tools/graph_bench.py [H W B]"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from polyblur_amd import _capi as capi
from polyblur_amd.engine import get_engine, Engine
from polyblur_amd.synthetic import synthetic_blurry_batch
H, W, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (500, 700, 1)
eng = get_engine(0)
xs, _ = synthetic_blurry_batch(B, 3, H, W, seed0=9)
x = torch.from_numpy(xs).cuda(); out = torch.empty_like(x)
o = Engine.make_options(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
for mode in (False, True):
    eng.set_graph_mode(mode)
    for _ in range(5): eng.polyblur_ptr(x.data_ptr(), out.data_ptr(), capi.PB_F32, x.shape, o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    for _ in range(n): eng.polyblur_ptr(x.data_ptr(), out.data_ptr(), capi.PB_F32, x.shape, o)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print("%dx%dx%d graph=%d : %.4f ms/call  %.1f MP/s" % (B, H, W, mode, ms, B * H * W / 1e3 / ms))
eng.set_graph_mode(False)
