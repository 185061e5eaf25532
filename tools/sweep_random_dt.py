"""The domain-transform filter on random shapes (python tools/sweep_random_dt.py [first last]): the default engine (register rows,
strips / stored weights / workgroups of columns as the library chooses), the same with the few-columns and the four-waves-per-row forms forced, one wave per row and thread per column, and the forms
through global memory -- the same bits -- against the oracle; and the forced few-columns form twenty times over on one input."""
import os, sys, numpy as np
sys.path.insert(0, '.')
from oracle import polyblur_ref as ref
from polyblur_amd.engine import Engine


def engine(**env):
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Engine(0)
    finally:
        for k in env: del os.environ[k]


a, b = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (0, 150)
engs = {"default": engine(), "coop": engine(PB_DT_COLS_COOP=2, PB_DT_ROWS_REG=3), "wave": engine(PB_DT_ROWS_REG=2, PB_DT_COLS_COOP=0), "global": engine(PB_DT_COLS_COOP=0, PB_DT_COLS_STRIP=0, PB_DT_ROWS_REG=0)}
bad = 0; worst = {np.float32: 0.0, np.float16: 0.0}
for i in range(a, b):
    rng = np.random.default_rng(9000 + i)
    B, C = int(rng.integers(1, 5)), int(rng.choice([1, 3]))
    H = int(rng.choice([rng.integers(2, 40), rng.integers(40, 500), 96 * int(rng.integers(1, 5)), 96 * int(rng.integers(1, 5)) + 1]))
    W = int(rng.choice([rng.integers(2, 80), rng.integers(2000, 8300), 4 * rng.integers(1, 200), 8 * rng.integers(1, 120), 16 * rng.integers(1, 80), 64 * rng.integers(1, 20), rng.integers(80, 900)]))
    dt = rng.choice([np.float32, np.float16])
    N = int(rng.integers(1, 4))
    ss, sr = float(rng.uniform(1.0, 60.0)), float(rng.uniform(0.1, 1.0))
    x = rng.random((B, C, H, W), dtype=np.float32).astype(dt)
    jt = rng.random((B, C, H, W), dtype=np.float32).astype(dt) if rng.random() < 0.4 else None
    outs = {k: e.dt_recursive_filter(x, ss, sr, N, joint=jt) for k, e in engs.items()}
    want = ref.recursive_filter(x.astype(np.float32), ss, sr, N, None if jt is None else jt.astype(np.float32))
    err = float(np.abs(outs["default"].astype(np.float32) - want).max())
    tol = 2e-5 if dt == np.float32 else 1e-3
    same = all(np.array_equal(outs["global"], outs[k]) for k in ("default", "coop", "wave"))
    worst[dt] = max(worst[dt], err)
    if not same or err >= tol:
        bad += 1
        print("case", i, (B, C, H, W), np.dtype(dt).name, N, round(ss, 2), round(sr, 2), "joint" if jt is not None else "", "same bits", same, "err %.3e" % err)
print("dt cases %d..%d: %d outside tolerance or differing between forms; worst error fp32 %.3e, fp16 I/O %.3e" % (a, b, bad, worst[np.float32], worst[np.float16]))
rng = np.random.default_rng(5)
for shape in ((2, 3, 1000, 1936), (1, 1, 777, 1280), (5, 3, 193, 640)):
    x = rng.random(shape, dtype=np.float32)
    first = engs["coop"].dt_recursive_filter(x, 8.0, 0.4, 3)
    n = sum(not np.array_equal(first, engs["coop"].dt_recursive_filter(x, 8.0, 0.4, 3)) for _ in range(20))
    print("few-columns form,", shape, "20 repeats:", n, "differ; equal to the global forms:", bool(np.array_equal(first, engs["global"].dt_recursive_filter(x, 8.0, 0.4, 3))))
