import sys, numpy as np, torch
sys.path.insert(0, ".")
from polyblur_amd import _capi as capi
from polyblur_amd.engine import get_engine
eng = get_engine(0)
x4 = torch.rand(1, 3, 2160, 3840, device="cuda"); o4 = torch.empty_like(x4)
eng.set_stream(torch.cuda.current_stream(0).cuda_stream)
for deg, sg, rh in ((0.0, 0.3, 0.3), (0.0, 0.6, 0.3), (0.0, 1.0, 0.6), (0.0, 2.0, 1.0), (30.0, 0.6, 0.3)):
    for support, sname in ((capi.PB_SUPPORT_ADAPTIVE, "adaptive"), (capi.PB_SUPPORT_FULL, "full")):
        buf = eng.make_kernels(np.array([sg], np.float32), np.array([rh], np.float32), np.array([np.deg2rad(deg)], np.float32), support=support, name="p1.info")
        rec = eng.read_info(buf, 1)
        ms = eng.time_inner_loop(x4.data_ptr(), o4.data_ptr(), capi.PB_F32, x4.shape, buf.ptr, 6.0, 1.0, capi.PB_WRAP, 20)
        print("theta %3.0f sigma %.2f rho %.2f %-8s radius %2d separable %d: 4K polynomial %.4f ms" % (deg, sg, rh, sname, int(rec["radius"][0]), int(rec["separable"][0]), ms), flush=True)
