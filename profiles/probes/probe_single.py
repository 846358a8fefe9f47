# latency of ONE 128x128 task (as on the factorisation's chain) per workgroup shape and K
import sys, os, numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
from gpyrn_amd import _hip
c=_hip.Context(0)
rng=np.random.RandomState(0)
M=N=128
for K in (128,512):
    A=rng.standard_normal((M,K)); B=rng.standard_normal((K,N)); C=rng.standard_normal((M,N))
    for shape in (0,1,2,3):
        best=1e9
        for rep in range(5):
            c.profile_enable()
            c.test_gemm(A,B,C,0,1,1|(shape<<4))
            pr=c.profile_read()
            best=min(best,pr['update'][0])
        print('K',K,'shape',['128x128','64x64','64x128','128x64'][shape],'us',round(best*1e3,1),flush=True)
