"""A few calls on the headline image with options given as key=value (for rocprofv3):
python tools/run_options.py remove_halo=True prefiltering=True prefilter=domain_transform [B=1 H=2160 W=3840 dtype=f16]"""
import os, sys, ast, torch
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch, DEFAULT_SEED
kw = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1)
geo = dict(B=1, H=2160, W=3840, dtype="f32")
for a in sys.argv[1:]:
    k, v = a.split("=")
    try: v = ast.literal_eval(v)
    except Exception: pass
    (geo if k in geo else kw)[k] = v
x = torch.from_numpy(synthetic_blurry_batch(min(geo["B"], 4), 3, geo["H"], geo["W"], seed0=DEFAULT_SEED)[0]).cuda()
x = x.repeat((geo["B"] + x.shape[0] - 1) // x.shape[0], 1, 1, 1)[:geo["B"]].contiguous()
if geo["dtype"] == "f16": x = x.half()
for _ in range(6): polyblur_deblurring(x, **kw)
torch.cuda.synchronize()
