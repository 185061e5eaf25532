"""A few cfg3-like calls (16 x 1080p fp16, halo masking + domain-transform prefilter) for rocprofv3: python tools/run_cfg3_small.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get('PB_PKG_ROOT', '.'))
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch, DEFAULT_SEED
x = torch.from_numpy(synthetic_blurry_batch(4, 3, 1080, 1920, seed0=DEFAULT_SEED)[0]).repeat(4, 1, 1, 1).cuda().half()
kw = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1, remove_halo=True, prefiltering=True, prefilter="domain_transform")
for _ in range(4): polyblur_deblurring(x, **kw)
torch.cuda.synchronize()
