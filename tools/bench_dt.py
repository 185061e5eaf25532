"""The domain-transform filter on resident tensors: the column pass by workgroups of columns (coop), as the library chooses
(default), one thread per column with PB_DT_COLS_STRIP = 1 (weights stored where they pay) / 2 (J again) / 0 (two sweeps): time
per call and the same bits as the two sweeps.  python tools/bench_dt.py [B H W f32|f16 N]"""
import os, sys, json, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from polyblur_amd.engine import Engine, _DT
B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 1080, 1920)
dt = np.float16 if len(sys.argv) > 4 and sys.argv[4] == "f16" else np.float32
N = int(sys.argv[5]) if len(sys.argv) > 5 else 1
tdt = torch.float16 if dt == np.float16 else torch.float32
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.rand((B, 3, H, W), device="cuda", generator=g).to(tdt).contiguous()
res, outs = {}, {}
MODES = {"coop": dict(PB_DT_COLS_COOP=2), "rows4w": dict(PB_DT_ROWS_REG=3), "rows1w": dict(PB_DT_ROWS_REG=2), "default": {}, "1": dict(PB_DT_COLS_COOP=0, PB_DT_COLS_STRIP=1), "2": dict(PB_DT_COLS_COOP=0, PB_DT_COLS_STRIP=2),
         "0": dict(PB_DT_COLS_COOP=0, PB_DT_COLS_STRIP=0)}
for mode, env in MODES.items():
    os.environ.update({k: str(v) for k, v in env.items()})
    eng = Engine(0)
    for k in env: del os.environ[k]
    out = torch.empty((B, 3, H, W), device="cuda", dtype=tdt)
    def call():
        eng._check(eng.lib.pb_dt_recursive_filter(eng.ctx, ctypes.c_void_p(x.data_ptr()), None, ctypes.c_void_p(out.data_ptr()), _DT[np.dtype(dt)], B, 3, H, W, 2.0, 0.8, N))
    for _ in range(3): call()
    eng.synchronize(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); eng.synchronize()
        import time
        t0 = time.perf_counter(); call(); eng.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    res[mode] = round(float(np.median(ts)), 4)
    outs[mode] = out.clone()
print(json.dumps(dict(shape=[B, 3, H, W], dtype=str(np.dtype(dt)), N=N, ms=res,
                      same_bits={k: bool(torch.equal(outs[k], outs["0"])) for k in outs})))
