import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle import polyblur_ref as ref
from polyblur_amd import _capi as capi
from polyblur_amd.engine import get_engine
eng=get_engine(0)
th=np.float32(0)
k=ref.gaussian_kernel_2d([th],[2.5],[1.0])
buf=eng.make_kernels([2.5],[1.0],[th])
info=eng.read_info(buf,1)
print('kx sym', np.abs(info['kx'][0]-info['kx'][0][::-1]).max(), 'ky', np.abs(info['ky'][0]-info['ky'][0][::-1]).max())
for cx in (100,101,102,103):
    xp=np.zeros((1,1,174,234),np.float32); xp[0,0,80,cx]=1
    out=eng.convolve2d(xp,buf,capi.PB_ZERO)
    want=ref.correlate_same_zero(xp,k[:,None])
    err=np.abs(out-want)[0,0]
    print('impulse col',cx,'max err',err.max())
    ys,xs=np.where(err>1e-7)
    if len(ys): print('  rows',ys.min(),ys.max(),'cols',xs.min(),xs.max(), 'n',len(ys))
    # effective vertical profile at column cx
    print('  got ', np.round(out[0,0,74:87,cx]*1e3,3))
    print('  want', np.round(want[0,0,74:87,cx]*1e3,3))
