import sys, numpy as np
sys.path.insert(0,'.')
from gpyrn_amd import _hip
c=_hip.Context(0)
rng=np.random.RandomState(0)
n=2048
t=np.sort(rng.uniform(0,0.4*n,n)); r=t[:,None]-t[None,:]
A=np.exp(-0.5*r**2/900)+np.eye(n)
for rep in range(2):
    c.profile_enable()
    c.test_factor_invert(A)
    pr=c.profile_read()
print({k:(round(v[0]*1e3/max(v[1],1),1),v[1]) for k,v in pr.items() if v[1]})
