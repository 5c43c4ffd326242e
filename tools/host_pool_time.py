import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np
from mimo_amd import _lib
from mimo_amd.distributions import StackedNormalWisharts, StackedMatrixNormalWisharts, composite
from mimo_amd.utils.abstraction import Statistics as Stats
def T(fn,n=300):
    fn(); best=1e9
    for r in range(9):
        t=time.perf_counter()
        for _ in range(n): fn()
        best=min(best,(time.perf_counter()-t)/n*1e6)
    return best
for K,D in ((64,16),(256,8),(128,32),(4,2),(16,16),(64,8)):
    rng=np.random.default_rng(0)
    A=rng.standard_normal((K,D,D)); kap=rng.uniform(0.5,200.,K); mus=rng.standard_normal((K,D))
    nat=[kap[:,None]*mus,kap,A@A.transpose(0,2,1)+D*np.eye(D)+kap[:,None,None]*np.einsum('kd,kl->kdl',mus,mus),rng.uniform(1.,5000.,K)]
    p=StackedNormalWisharts(K,D)
    print("NW K=%d D=%d: assign %.1f us"%(K,D,T(lambda: p._assign_native(nat))))
