import sys, numpy as np
sys.path.insert(0,'.')
from gpyrn_amd import _hip
c=_hip.Context(0)
rng=np.random.RandomState(0)
for (M,N,K) in [(4096,4096,128),(4096,4096,512),(4096,4096,2048),(8192,8192,512)]:
    A=rng.standard_normal((M,K)); B=rng.standard_normal((K,N)); C=rng.standard_normal((M,N))
    for modes in [(0,0,1),(0,1,1)]:
        for rep in range(2):
            c.profile_enable()
            c.test_gemm(A,B,C,modes[0],modes[1],modes[2])
            pr=c.profile_read()
        ms=pr['update'][0]
        print(M,N,K,modes,'ms',round(ms,3),'TF',round(2*M*N*K/ms/1e9,1))
