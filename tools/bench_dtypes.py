"""ms per call of the default pipeline on the same image as fp32, fp16 and 8-bit planes (run on the GPU box)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from polyblur_amd import polyblur_deblurring, polyblur_deblurring_uint8
from polyblur_amd.synthetic import synthetic_blurry_batch
KW = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1)
for (b, c, h, w) in ((1, 3, 2160, 3840), (8, 3, 1080, 1920), (1, 3, 500, 700)):
    x32 = torch.from_numpy(synthetic_blurry_batch(b, c, h, w, seed0=7)[0]).cuda()
    for name, x in (("f32", x32), ("f16", x32.half()), ("u8", (x32 * 255).round().clamp(0, 255).to(torch.uint8))):
        f = polyblur_deblurring_uint8 if name == 'u8' else polyblur_deblurring
        for _ in range(5): f(x, **KW)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): f(x, **KW)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 30 * 1e3
        print("%dx%dx%dx%d %s %.3f ms/call" % (b, c, h, w, name, ms), flush=True)
