# which tiles of L / X are wrong with the persistent chain kernel?
import sys, numpy as np
sys.path.insert(0,'.')
from gpyrn_amd import _hip
def spd(n, rng, shift=1.0):
    t = np.sort(rng.uniform(0, 0.4 * n, n)); r = t[:, None] - t[None, :]
    return np.exp(-0.5 * r**2 / 30.0**2) + shift * np.eye(n)
c=_hip.Context(0)
for n,batch in ((384,1),(512,1),(640,1),(640,3),(1024,2),(2048,2)):
    rng=np.random.RandomState(n)
    A=np.array([spd(n,rng,1.0+b) for b in range(batch)])
    for rep in range(3):
        L,X,info=c.test_factor_invert(A)
        T=n//128
        bad=[]
        for b in range(batch):
            Lr=np.linalg.cholesky(A[b]); Xr=np.linalg.inv(Lr)
            for i in range(T):
                for j in range(i+1):
                    eL=np.abs(L[b][i*128:(i+1)*128,j*128:(j+1)*128]-Lr[i*128:(i+1)*128,j*128:(j+1)*128]).max()
                    eX=np.abs(np.tril(X[b])[i*128:(i+1)*128,j*128:(j+1)*128]-Xr[i*128:(i+1)*128,j*128:(j+1)*128]).max()
                    if eL>1e-10 or eX>1e-9: bad.append((b,i,j,float('%.1e'%eL),float('%.1e'%eX)))
        print('n',n,'batch',batch,'rep',rep,'info',info,'bad tiles',bad[:12],len(bad),flush=True)
