import sys, numpy as np
sys.path.insert(0,'.')
from gpyrn_amd import _hip
c=_hip.Context(0)
rng=np.random.RandomState(0)
n=128
t=np.sort(rng.uniform(0,0.4*n,n)); r=t[:,None]-t[None,:]
A=np.exp(-0.5*r**2/900)+np.eye(n)
for rep in range(3):
    L,X,info=c.test_factor_invert(A)
    print('stamps (cycles from start):', X[0][0,16:16+5])
