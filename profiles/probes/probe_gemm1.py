import sys, numpy as np
sys.path.insert(0,'.')
from gpyrn_amd import _hip
c=_hip.Context(0)
rng=np.random.RandomState(0)
M,N,K=4096,4096,2048
A=rng.standard_normal((M,K)); B=rng.standard_normal((K,N)); C=rng.standard_normal((M,N))
for rep in range(3):
    c.test_gemm(A,B,C,0,0,1)
