import os, sys, numpy as np
sys.path.insert(0, '.')
from polyblur_amd.engine import Engine
from polyblur_amd.synthetic import synthetic_blurry_batch
def eng(**env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Engine(0)
    finally:
        for k, v in old.items():
            if v is None: del os.environ[k]
            else: os.environ[k] = v
a, b = eng(PB_COLS_FIXED=0), eng()
for shape in [(4,3,1080,1920), (8,1,1080,1920), (4,3,1080,256), (1,3,2160,3840), (1,3,4320,7680)]:
    B,C,H,W = shape
    img,_ = synthetic_blurry_batch(min(B,2),C,H,W,seed0=41)
    img = np.concatenate([img]*B)[:B]
    o = Engine.make_options(c=0.362,b=0.468)
    ra, rb = a.estimate_blur(img,o), b.estimate_blur(img,o)
    ra2 = a.estimate_blur(img,o)
    for f in ("mags","theta","sigma","rho","gray_min","gray_max"):
        x, y, z = np.asarray(ra[f]), np.asarray(rb[f]), np.asarray(ra2[f])
        print(shape, f, "equal" if np.array_equal(x,y) else "DIFF max %.3e" % np.max(np.abs(x-y)), "| generic twice:", np.array_equal(x, z))
    print(np.asarray(ra["mags"])[:2,:7]); print(np.asarray(rb["mags"])[:2,:7])
