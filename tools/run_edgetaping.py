"""A few calls with edgetaping on the headline image (for rocprofv3): python tools/run_edgetaping.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch, DEFAULT_SEED
x = torch.from_numpy(synthetic_blurry_batch(1, 3, 2160, 3840, seed0=DEFAULT_SEED)[0]).cuda()
kw = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1, edgetaping=True)
for _ in range(6): polyblur_deblurring(x, **kw)
torch.cuda.synchronize()
