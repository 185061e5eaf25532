"""Same-box A/B of one configuration between two builds: PB_PKG_ROOT=<tree> python tools/ab_config.py B H W f32|f16 [n_iter]"""
import os, sys, json, time, numpy as np, torch
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import polyblur_deblurring
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch, DEFAULT_SEED
B, H, W = (int(v) for v in sys.argv[1:4])
dt = torch.float16 if sys.argv[4] == "f16" else torch.float32
n_iter = int(sys.argv[5]) if len(sys.argv) > 5 else 3
nd = min(B, 4)
x = synthetic_blurry_batch(nd, 3, H, W, seed0=DEFAULT_SEED)[0]
x = np.concatenate([x] * ((B + nd - 1) // nd))[:B]
x = torch.from_numpy(x).cuda().to(dt).contiguous()
kw = dict(n_iter=n_iter, c=0.362, b=0.468, alpha=6, beta=1)
eng = get_engine(0)
for _ in range(2): polyblur_deblurring(x, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): polyblur_deblurring(x, **kw)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
_, infos = polyblur_deblurring(x, return_info=True, **kw)
print(json.dumps(dict(ms=round(ms, 3), mp_per_s=round(B * H * W / 1e3 / ms, 1),
                      radius=[[int(r) for r in i["radius"][:4]] for i in infos], sep=[[int(r) for r in i["separable"][:4]] for i in infos])))
