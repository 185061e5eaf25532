import numpy as np, sys
sys.path.insert(0,'/root/repo')
from polyblur_amd.engine import get_engine, Engine
from polyblur_amd.synthetic import synthetic_blurry_batch
eng=get_engine(0)
x,_=synthetic_blurry_batch(2,3,160,224,seed0=55)
q=1e-4
info=eng.estimate_blur(x, Engine.make_options(c=0.362,b=0.468,q=q))
gray=x.mean(axis=1,dtype=np.float32)
flat=np.sort(gray.reshape(2,-1),axis=1)
n1=np.float32(flat.shape[1]-1)
r=np.float32(1.0-q)*n1; f=int(np.floor(r)); print('r',r,'f',f,'w',r-f)
print('dev hi', info['gray_max'], 'dev lo', info['gray_min'])
for b in range(2):
    print('sorted around', flat[b,f-2:f+3], 'max', flat[b,-1])
    d=info['gray_max'][b]; print(' position of dev value', np.searchsorted(flat[b], d))
