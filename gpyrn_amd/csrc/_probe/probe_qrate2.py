import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from gpyrn_amd import _hip
ctx = _hip.Context(0)
for (M, N, K) in [(8192, 8192, 128), (8192, 8192, 512)]:
    for how in (2, 3):
        print(f'{M}x{N}x{K} how={how}: {ctx.gemm_rate(M, N, K, how):6.1f} TF', flush=True)
