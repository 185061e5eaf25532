"""Per-stage hipEvent breakdown of one configuration: python tools/stage_profile.py B H W dtype [halo] [dt|bilateral|nc] [edgetaper]"""
import sys, json, numpy as np, torch
sys.path.insert(0, '.')
from polyblur_amd import polyblur_deblurring
from polyblur_amd.engine import get_engine
from polyblur_amd.synthetic import synthetic_blurry_batch
B, H, W = (int(v) for v in sys.argv[1:4])
dt = torch.float16 if sys.argv[4] == "f16" else torch.float32
flags = sys.argv[5:]
kw = dict(n_iter=3, c=0.362, b=0.468, alpha=6, beta=1)
if "halo" in flags: kw["remove_halo"] = True
if "edgetaper" in flags: kw["edgetaping"] = True
for name, key in (("dt", "domain_transform"), ("bilateral", "bilateral"), ("nc", "normalized_convolution")):
    if name in flags: kw.update(prefiltering=True, prefilter=key)
small, _ = synthetic_blurry_batch(min(B, 4), 3, H // 8, W // 8, seed0=5)
x = torch.nn.functional.interpolate(torch.from_numpy(small).cuda(), size=(H, W), mode="bicubic", align_corners=False).clamp(0, 1)
x = torch.cat([x] * ((B + 3) // 4))[:B].to(dt).contiguous()
eng = get_engine(0)
for _ in range(2): polyblur_deblurring(x, **kw)
torch.cuda.synchronize()
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.profile_begin()
import time
t0 = time.perf_counter()
for _ in range(3): polyblur_deblurring(x, **kw)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
prof = eng.profile_end()
print(json.dumps(dict(ms=round(ms, 3), mp_per_s=round(B * H * W / 1e3 / ms, 1), stages_ms={k: (round(v[0] / 3, 3), v[1] // 3) for k, v in prof.items() if v[1]})))
