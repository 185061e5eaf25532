"""Lab: the few-columns form of the column pass alone (PB_DT_COLS_COOP=2) -- per-call time of the filter, N = 1, device events.
python tools/bench_dt_coop.py B H W"""
import os, sys, json, ctypes, numpy as np, torch
sys.path.insert(0, '.')
os.environ["PB_DT_COLS_COOP"] = "2"
from polyblur_amd.engine import Engine, _DT
B, H, W = (int(v) for v in sys.argv[1:4])
x = torch.rand((B, 3, H, W), device="cuda").contiguous()
out = torch.empty_like(x)
eng = Engine(0)
def call():
    eng._check(eng.lib.pb_dt_recursive_filter(eng.ctx, ctypes.c_void_p(x.data_ptr()), None, ctypes.c_void_p(out.data_ptr()), _DT[np.dtype(np.float32)], B, 3, H, W, 2.0, 0.8, 1))
for _ in range(3): call()
eng.synchronize()
import time
ts = []
for _ in range(20):
    eng.synchronize(); t0 = time.perf_counter(); call(); eng.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
print(os.environ.get("POLYBLUR_HIP_LIB", "default"), [B, H, W], "us per call: median %.1f min %.1f" % (np.median(ts), min(ts)))
