import sys; sys.path.insert(0,'.')
from gpyrn_amd import _hip
c=_hip.Context(0)
print('flags', c.option('flags'), 'chain_streams', c.option('chain_streams'))
