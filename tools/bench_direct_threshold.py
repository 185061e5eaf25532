import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from polyblur_amd import polyblur_deblurring
from polyblur_amd.synthetic import synthetic_blurry_batch
KW = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1, method="direct")
for (b, c, h, w) in ((1, 3, 1080, 1920), (1, 3, 1440, 2560), (1, 3, 2000, 3000), (1, 3, 2160, 3840), (1, 3, 4320, 7680), (4, 3, 2160, 3840)):
    x = torch.from_numpy(synthetic_blurry_batch(1, c, h, w, seed0=7)[0]).repeat(b, 1, 1, 1).cuda()
    for _ in range(3): polyblur_deblurring(x, **KW)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(15): polyblur_deblurring(x, **KW)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 15 * 1e3
    print("direct %dx%dx%dx%d %.3f ms/call" % (b, c, h, w, ms))
