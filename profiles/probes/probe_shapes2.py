# stand-alone throughput of k_tile_gemm per workgroup shape and K on a 4096-task launch
# (c_mode bits 4-5: 0=128x128 (8 waves), 1=64x64, 2=64x128, 3=128x64)
import sys, numpy as np
sys.path.insert(0,'.')
from gpyrn_amd import _hip
c=_hip.Context(0)
rng=np.random.RandomState(0)
M=N=8192
for K in (128,512,2048):
    A=rng.standard_normal((M,K)); B=rng.standard_normal((K,N)); C=rng.standard_normal((M,N))
    for b_mode in (0,1):
        for shape in (0,1):
            for rep in range(2):
                c.profile_enable()
                c.test_gemm(A,B,C,0,b_mode,1|(shape<<4))
                pr=c.profile_read()
            ms=pr['update'][0]
            print('K',K,'b_mode',b_mode,'shape',['128x128','64x64','64x128','128x64'][shape],'ms',round(ms,3),'TF',round(2*M*N*K/ms/1e9,1),flush=True)
print('mfma peak', c.mfma_peak(2,4000))
