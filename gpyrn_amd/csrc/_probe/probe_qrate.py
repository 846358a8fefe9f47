"""Tile contraction rate: one launch vs the dataflow schedule's worker kernel on the same independent tasks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from gpyrn_amd import _hip
ctx = _hip.Context(0)
names = {0: 'launch 64x64', 1: 'launch 128x128', 2: 'queue quarters', 3: 'queue whole nodes'}
for (M, N, K) in [(8192, 8192, 512), (8192, 8192, 128), (2048, 2048, 128), (1024, 1024, 128)]:
    for how in (0, 1, 2, 3):
        print(f'{M}x{N}x{K} {names[how]:18s}: {ctx.gemm_rate(M, N, K, how):6.1f} TF', flush=True)
