"""Lab: the 5x5 bilateral prefilter on resident tensors: python tools/bench_bilateral.py B H W [f16]"""
import sys, ctypes, time, numpy as np, torch
sys.path.insert(0, '.')
from polyblur_amd.engine import Engine, _DT
from oracle import polyblur_ref as ref
B, H, W = (int(v) for v in sys.argv[1:4])
dt = np.float16 if len(sys.argv) > 4 and sys.argv[4] == "f16" else np.float32
x = torch.rand((B, 3, H, W), device="cuda").to(torch.float16 if dt == np.float16 else torch.float32).contiguous()
out = torch.empty_like(x)
eng = Engine(0)
def call():
    eng._check(eng.lib.pb_bilateral5(eng.ctx, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), _DT[np.dtype(dt)], B, 3, H, W))
for _ in range(3): call()
eng.synchronize()
ts = []
for _ in range(20):
    eng.synchronize(); t0 = time.perf_counter(); call(); eng.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
small = x[:1, :, :200, :300].float().cpu().numpy()
got = eng.bilateral5(small)
print([B, 3, H, W], np.dtype(dt).name, "us per call: median %.1f min %.1f; max |.| against the oracle on a 200 x 300 crop: %.2e" % (np.median(ts), min(ts), np.abs(got - ref.bilateral_filter(small)).max()))
