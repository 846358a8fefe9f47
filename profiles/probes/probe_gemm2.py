# one K=512 bulk-shaped launch set (run under rocprofv3 --pmc ...; GPRN_PAD_ALL=1 for one workgroup per CU)
import sys, numpy as np
import os; sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..')))
from gpyrn_amd import _hip
c=_hip.Context(0)
rng=np.random.RandomState(0)
M,N,K=8192,8192,512
A=rng.standard_normal((M,K)); B=rng.standard_normal((K,N)); C=rng.standard_normal((M,N))
for rep in range(3):
    c.test_gemm(A,B,C,0,1,1)
