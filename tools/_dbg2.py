import sys; sys.path.insert(0,'.')
import numpy as np
from tests.test_gpu_parity import _setup, DCASES
from tests.emul_engine import EmulEngine
case=DCASES[0]
s=_setup(case,2,seed=11); n,dim,hs,ht=case[:4]
v=s["rng"].standard_normal(s["x"].shape)
eng=s["flow"].engine(n,dim,s["sp"]); eng.set_params(s["theta"])
em=EmulEngine(n,dim,2,hs,ht,s["L"],s["sp"]); em.set_params(s["theta"])
ge,le=em.grad_laplacian(s["x"],s["sidx"],0,v)
for thr in (0,64,128,256,512,0,0):
    eng.set_block_threads(thr)
    g,l=eng.grad_laplacian(s["x"],s["sidx"],0,v)
    print('thr',thr,'grad diff',np.abs(g-ge).max(), 'per walker', np.abs(g-ge).max(axis=(1,2)), 'lap', np.abs(l-le))
# single-walker calls
for b in range(2):
    g,l=eng.grad_laplacian(s["x"][b],s["sidx"][b],0,v[b]); print('single',b,np.abs(g-ge[b]).max())
# x-perturbation sensitivity through the emulation: condition estimate
x2=s["x"]*(1+1e-12); g2,l2=em.grad_laplacian(x2,s["sidx"],0,v); print('emul sensitivity to 1e-12 rel x change:',np.abs(g2-ge).max())
