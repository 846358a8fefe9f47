import sys
sys.path.insert(0,'.')
from gpyrn_amd import _hip
c=_hip.Context(0)
for w in (1,2):
    for it in (2000, 20000):
        print('wg/cu',w,'iters',it,'TF', round(c.mfma_peak(w,it),2))
