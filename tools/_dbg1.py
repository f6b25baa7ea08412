import sys; sys.path.insert(0,'.')
import numpy as np
from tests.test_gpu_parity import _setup, DCASES, CASES
from tests.emul_engine import EmulEngine
import coulombgas_amd as cg
for case in [DCASES[0], DCASES[1], DCASES[2]]:
    s=_setup(case,2,seed=11); n,dim,hs,ht=case[:4]
    v=s["rng"].standard_normal(s["x"].shape)
    eng=s["flow"].engine(n,dim,s["sp"]); eng.set_params(s["theta"])
    em=EmulEngine(n,dim,2,hs,ht,s["L"],s["sp"]); em.set_params(s["theta"])
    for thr in (0,64):
        eng.set_block_threads(thr)
        for mode in (0,1,2):
            g,l=eng.grad_laplacian(s["x"],s["sidx"],mode,v); ge,le=em.grad_laplacian(s["x"],s["sidx"],mode,v)
            print(case[:4],'thr',thr,'mode',mode,'grad diff',np.abs(g-ge).max(),'lap diff',np.abs(l-le).max())
    print(' logpsi diff', np.abs(eng.logpsi(s["x"],s["sidx"])-em.logpsi(s["x"],s["sidx"])).max(), 'J diff', np.abs(eng.flow_jacobian(s["x"])-em.flow_jacobian(s["x"])).max())
