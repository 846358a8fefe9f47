# timeline of the persistent chain kernel (GPRN_CHAIN_STAMPS=1): one factor+inverse at N=2048 (1 matrix) and N=4096 (2, 6)
import sys, numpy as np
sys.path.insert(0,'.')
from gpyrn_amd import _hip
def spd(n, rng, shift=1.0):
    t = np.sort(rng.uniform(0, 0.4 * n, n)); r = t[:, None] - t[None, :]
    return np.exp(-0.5 * r**2 / 30.0**2) + shift * np.eye(n)
c=_hip.Context(0)
for n,batch in ((2048,1),(4096,2),(4096,6)):
    rng=np.random.RandomState(n)
    A1=spd(n,rng)
    A=np.array([A1 + 0.1*b*np.eye(n) for b in range(batch)])
    for rep in range(2):
        print('n',n,'batch',batch,flush=True)
        c.profile_enable(['diag'])
        L,X,info=c.test_factor_invert(A)
        print('  diag family ms', c.profile_read()['diag'], 'info', info, flush=True)
