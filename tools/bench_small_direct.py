import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch
KW = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1, method="direct")
for (b, c, h, w) in ((1, 3, 256, 256), (1, 3, 500, 700), (1, 3, 720, 1280), (1, 3, 1080, 1920), (16, 3, 500, 700)):
    x = torch.from_numpy(synthetic_blurry_batch(b, c, h, w, seed0=7)[0]).cuda()
    for _ in range(5): polyblur_deblurring(x, **KW)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): polyblur_deblurring(x, **KW)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
    print("direct %dx%dx%dx%d %.3f ms/call" % (b, c, h, w, ms))
