# round 6: stand-alone rate of the tile kernel, 64x64 (4 waves) / 128x128 (8 waves) / 128x128 on FOUR waves (how 0 / 1 / 2)
import sys, numpy as np
sys.path.insert(0, '.')
from gpyrn_amd import _hip
c = _hip.Context(0)
for K in (512, 128, 2048):
    for how in (0, 1, 2):
        best = max(c.gemm_rate(8192, 8192, K, how, reps=5) for _ in range(3))
        print('K %4d  %-22s %.1f TF' % (K, ['64x64 (4 waves)', '128x128 (8 waves)', '128x128 (4 waves)'][how], best), flush=True)
print('mfma peak', c.mfma_peak(2, 4000))
# correctness of the new shape: integer data, every layout
rng = np.random.RandomState(5)
for a_mode, b_mode in ((0, 0), (0, 1), (1, 1), (1, 0)):
    for c_mode in (0, 1, 2):
        A = rng.randint(-4, 5, size=(256, 80)).astype(float); B = rng.randint(-4, 5, size=(80, 384)).astype(float)
        C0 = rng.randint(-9, 10, size=(256, 384)).astype(float)
        out = c.test_gemm(A, B, C0, a_mode, b_mode, c_mode | (4 << 4))
        want = {0: A @ B, 1: C0 - A @ B, 2: -(A @ B)}[c_mode]
        assert np.array_equal(out, want), (a_mode, b_mode, c_mode)
print('layouts ok')
