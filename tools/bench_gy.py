import sys, time, numpy as np, torch
sys.path.insert(0,'.')
from polyblur_amd.engine import get_engine
eng=get_engine(0)
import ctypes as C
for shape in ((192,1080,1920),(3,2160,3840),(3,500,700)):
    P,H,W=shape
    x=torch.rand(P,H,W,device='cuda'); gy=torch.empty_like(x)
    f=lambda: eng._check(eng.lib.pb_fourier_gradients(eng.ctx, x.data_ptr(), P, H, W, None, gy.data_ptr()))
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); print(shape, "%.1f us"%((time.perf_counter()-t0)/10*1e6), float(gy.abs().sum()))
